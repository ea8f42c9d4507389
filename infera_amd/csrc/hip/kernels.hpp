// kernels.hpp -- host-callable launchers for the gfx950 kernels.  Every launcher enqueues on the
// given stream and returns immediately; errors surface through hipGetLastError at the call site.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

namespace infera_hip::kern {

// Activation codes shared with plan.hpp (Act enum values).
struct ActParam {
  int kind = 0;  // plan.hpp Act: 0 none, 1 relu, 2 sigmoid, 3 tanh, 4 leaky-relu(a), 5 clip(a,b), 6.. elementwise-only kinds
  float a = 0.f, b = 0.f;
};

// ---- elementwise / reductions (eltwise.hip) --------------------------------------------------
void unary(hipStream_t s, const float *x, float *y, int64_t n, ActParam act);
// y[r, i] = act(x[r, i] (op) c[i])   (const_left: c (op) x)
void binary_const(hipStream_t s, const float *x, const float *c, float *y, int64_t rows, int64_t per_row, char op,
                  bool const_left, ActParam act);
void binary_act(hipStream_t s, const float *a, const float *b, float *y, int64_t n, char op, ActParam act);
// y[r, c, i] = act(a[r, c, i] (op) gate[r, c]): per-channel gate broadcast over the S positions of a channel
void binary_gate(hipStream_t s, const float *a, const float *gate, float *y, int64_t rows, int64_t C, int64_t S, char op, ActParam act,
                 bool cq);
// y[r, c, i] = act(x * scale[c] + shift[c]);  cq: activations in channel-quad planes [N][C/4][S][4] (conv.hip)
void affine_channel(hipStream_t s, const float *x, const float *scale, const float *shift, float *y, int64_t rows,
                    int64_t C, int64_t S, ActParam act, bool cq);
// row reduction + rescale over `len` with element stride `inner`, repeated rows*outer*inner times.
// mode 0 softmax, 1 log-softmax; 2 / 3 / 4: divide by max|x| / sum|x| / sqrt(sum x^2) (ai.onnx.ml Normalizer, divisor floored at 1e-30)
void softmax(hipStream_t s, const float *x, float *y, int64_t rows, int64_t outer, int64_t len, int64_t inner, int mode);
// dst[r, dst_off : dst_off+len] = src[r, src_off : src_off+len]  (Concat piece / Slice / Split); y[r] = float(argmax_j x[r, j])
void copy_cols(hipStream_t s, const float *src, float *dst, int64_t rows, int64_t len, int64_t src_stride, int64_t src_off,
               int64_t dst_stride, int64_t dst_off);
void argmax_rows(hipStream_t s, const float *x, float *y, int64_t rows, int64_t len);
// dst[r, 0:K] = src[r, :], zeros up to Kp columns
void pad_cols(hipStream_t s, const float *src, float *dst, int64_t rows, int64_t K, int64_t Kp);
// dst[rows][ncols] = transpose of src[ncols][rows]  (column-major staging of a columnar chunk)
void transpose_cm(hipStream_t s, const float *src, float *dst, int64_t rows, int64_t ncols);
// Zero-copy column gather: up to kMaxZeroCopyCols device-visible column-run pointers (registered host memory) -> [ncols][rows] f32
// in HBM.  type: 0 f32, 1 f64, 2 i32, 3 i64 (+8 = constant vector: one element); passed by value as the kernel argument (2.3 KB).
constexpr int kMaxZeroCopyCols = 256;
struct ColumnTable {
  const void *ptr[kMaxZeroCopyCols];
  unsigned char type[kMaxZeroCopyCols];
};
void gather_columns_device(hipStream_t s, const ColumnTable &tab, int ncols, int64_t rows, float *dst);
// synthetic table fill (SURVEY.md 8d generator), row-major [rows, ncols]
void synth_fill(hipStream_t s, float *dst, uint64_t seed, uint64_t row0, uint64_t rows, uint64_t ncols);

// ---- dense layer, fp32 MFMA (dense.hip) -------------------------------------------------------
// Y[rows, M] = act(X[rows, K] . W[K, M] + bias[M]); W row-major, bias may be null.
// softmax_fused: apply a row softmax over the M outputs in the epilogue (requires M <= 256).
// x_colmajor: X is ONE column-major chunk [K][rows] (host path staging; only where dense_colmajor_supported(K, M)).
// kernel family that serves a Dense layer (the launcher's own decision function; "" = none carries that epilogue)
const char *dense_kernel_family(int64_t rows, int K, int M, int softmax_mode, bool x_colmajor, bool aligned16);
void dense(hipStream_t s, const float *X, const float *W, const float *bias, float *Y, int64_t rows, int K, int M,
           ActParam act, int softmax_mode /*0 none,1 softmax,2 log-softmax,3 argmax (label only)*/, bool x_colmajor = false);
bool dense_colmajor_supported(int K, int M);
// where a streaming kernel carries the softmax epilogue (wider heads: tiled Dense + the row softmax kernel is faster)
bool dense_can_fuse_softmax(int K, int M);
// softmax_mode 3: Y[rows] = float(index of the first maximum of the M scores) -- only where this returns true
bool dense_can_fuse_argmax(const float *X, int K, int M);

// ---- whole-chain fused MLP (mlp_fused.hip) -----------------------------------------------------
// A chain D0 -> D1 -> D2 -> D3 evaluated in one persistent kernel; activations never leave
// registers.  `packed` holds the fragment-major weights produced by mlp3_pack().
struct Mlp3Shape {
  int d0, d1, d2, d3;
  int act1, act2, act3;  // only None/Relu chains are instantiated ahead of time
};
// True if a fused kernel exists for the chain: an ahead-of-time instantiation, or one compiled right now
// by hipRTC from the same device source (needs a visible GPU).  On false, `why` explains and the caller
// keeps the layer-by-layer plan.
bool mlp3_supported(const Mlp3Shape &sh, std::string *why = nullptr);
// Size in floats of the packed weight blob and host-side packing (fragment-major order, see mlp_fused.hip).
size_t mlp3_packed_floats(const Mlp3Shape &sh);
void mlp3_pack(const Mlp3Shape &sh, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
               const float *b3, float *packed);
// x_colmajor: X is a column-major chunk [d0][rows] (the host path's staging of flat DuckDB columns); only for the chains
// mlp3_colmajor_supported() names.
bool mlp3(hipStream_t s, const Mlp3Shape &sh, const float *X, const float *packed, float *Y, int64_t rows, int num_cus,
          std::string *why = nullptr, bool x_colmajor = false);
bool mlp3_colmajor_supported(const Mlp3Shape &sh);
int64_t mlp3_colmajor_max_rows(const Mlp3Shape &sh);  // longest column-major chunk the chain's kernels read themselves (0: none)
std::string mlp3_kernel_name(const Mlp3Shape &sh);

// ---- fused chain of small Dense layers over tables of any width (chain_device.inc, specialised with hipRTC) ----
// k0 table columns; layer l maps dims[l-1] (dims[-1] = k0) -> dims[l] and applies acts[l] (plan.hpp Act 0..5) with
// parameters pa/pb; sm: 0 plain, 1 softmax, 2 log-softmax, 3 argmax (label only) over the last layer's <= 16 outputs.
struct ChainShape {
  int k0 = 0;
  std::vector<int> dims, acts;
  std::vector<float> pa, pb;
  int sm = 0;
};
// True once a kernel for the shape is compiled (first call compiles); `why` otherwise: limits, hipRTC missing, ...
bool chain_supported(const ChainShape &s, std::string *why = nullptr);
size_t chain_packed_floats(const ChainShape &s);
// W[l] is [K_l, M_l] row-major, bias[l] has M_l floats or is null
void chain_pack(const ChainShape &s, const std::vector<const float *> &W, const std::vector<const float *> &bias, float *out);
// x_colmajor: X is one column-major chunk [k0][rows] (the host path's staging layout) instead of the row-major table
bool chain(hipStream_t st, const ChainShape &s, const float *X, const float *packed, float *Y, int64_t rows, int num_cus,
           std::string *why, bool x_colmajor = false);
std::string chain_kernel_name(const ChainShape &s);

// ---- convolution / pooling (conv.hip) ----------------------------------------------------------
struct ConvGeom {
  int C, H, W, M, OH, OW, kh, kw, sh, sw, pt, pl, dh, dw, groups;
  // Dense layers on the tiled kernel (H = W = 1): C and M are the PADDED sizes (multiples of 32, zero weights / bias
  // beyond the real ones); the real row lengths of the input and output matrices.  0 = not a dense layer.
  int kvalid = 0, mvalid = 0;
  // Convolutions whose channel counts are whole quads but no multiples of 32 (MobileNet's 16 / 24 / 144, ...): C and M
  // are the PADDED sizes as above, kvalid / mvalid the real ones (the tensors' plane counts are kvalid/4 and mvalid/4).
  int padc = 0;
};
// the geometry the tiled kernel runs for a convolution step: channels padded to 32 when they are not (padc = 1)
ConvGeom conv2d_tiled_geom(const ConvGeom &real);
// the same for the stem's patch kernel: M padded to whole 32-feature tiles (mvalid = the real count) for stems with 16 / 24 outputs
ConvGeom conv2d_patch_geom(const ConvGeom &real);
// Generic implicit-GEMM convolution: any geometry / groups; Wk = conv2d_generic_pack() of the ONNX
// weights ([group][k][M/g], k = (c, ky, kx)); activations NCHW or channel-quad planes (CQ) per flag.
bool conv2d_generic_supported(const ConvGeom &g);  // (C/g)*kh*kw <= 8192 (the per-k offset table lives in LDS)
void conv2d_generic_pack(const ConvGeom &g, const float *Wt, float *packed);
void conv2d(hipStream_t s, const float *X, const float *Wk, const float *bias, float *Y, int64_t rows, const ConvGeom &g,
            ActParam act, bool in_cq, bool out_cq);
// Patch convolution for the network's first layer: few input channels (C <= 8), NCHW input, CQ output,
// M in {32, 64, 96, 128}, groups == 1; the receptive field of a pixel tile is staged in LDS.
bool conv2d_patch_supported(const ConvGeom &g);
size_t conv2d_patch_packed_floats(const ConvGeom &g);
// ... with the MaxPool 3x3 / stride 2 that follows it in the same kernel (ResNet / DenseNet / SqueezeNet stems): the convolution's
// own output never reaches memory.  pool = pooled extent and the pooling's pads (0 or 1).  Pack with the same PoolTail.
struct PoolTail {
  int OH = 0, OW = 0, pt = 0, pl = 0;
};
bool conv2d_patch_pool_supported(const ConvGeom &g, const PoolTail &pool);
// ... and in the default arithmetic (bf16 x three exact parts, six MFMAs per product: no scales, nothing to track)
bool conv2d_stem_split6_supported(const ConvGeom &g, const PoolTail &pool);
size_t conv2d_stem_split6_packed_floats();
void conv2d_stem_split6_pack(const ConvGeom &g, const float *Wt, float *packed);
void conv2d_stem_split6(hipStream_t s, const float *X, const float *packed, const float *bias, float *Y, int64_t rows, const ConvGeom &g,
                        ActParam act, const PoolTail &pool, int num_cus);
void conv2d_patch_pool(hipStream_t s, const float *X, const float *packed, const float *bias, float *Y, int64_t rows,
                       const ConvGeom &g, ActParam act, const PoolTail &pool, int num_cus);
void conv2d_patch_pack(const ConvGeom &g, const float *Wt, float *packed, const PoolTail *pool = nullptr);
void conv2d_patch(hipStream_t s, const float *X, const float *packed, const float *bias, float *Y, int64_t rows,
                  const ConvGeom &g, ActParam act, int num_cus);
// Depthwise convolution (groups == C == M, C % 4 == 0) in channel-quad planes; packed = [C/4][tap][4].
bool conv2d_depthwise_supported(const ConvGeom &g);
void conv2d_depthwise_pack(const ConvGeom &g, const float *Wt, float *packed);
void conv2d_depthwise(hipStream_t s, const float *X, const float *packed, const float *bias, float *Y, int64_t rows,
                      const ConvGeom &g, ActParam act);
// Tiled CQ-layout convolution (groups == 1, C % 32 == 0, M % 32 == 0) on fragment-major packed weights.
bool conv2d_tiled_supported(const ConvGeom &g);
size_t conv2d_tiled_packed_floats(const ConvGeom &g);
void conv2d_tiled_pack(const ConvGeom &g, const float *Wt, float *packed);
// residual (nullable): CQ tensor of the output's shape added before the activation (fused ResNet Add)
void conv2d_tiled(hipStream_t s, const float *X, const float *packed, const float *bias, const float *residual, float *Y,
                  int64_t rows, const ConvGeom &g, ActParam act);
// The same convolution on the bf16 matrix cores with every fp32 operand cut exactly into three bf16 parts, six partial products per product (the default;
// conv_split.hip): no scales, no maxima,
// no precondition on the data.  M % 64 == 0; packed = conv2d_split6_packed_floats(g) floats' worth of bf16 hi / mid / lo fragments.
bool conv2d_split6_supported(const ConvGeom &g);
size_t conv2d_split6_packed_floats(const ConvGeom &g);
void conv2d_split6_pack(const ConvGeom &g, const float *Wt, float *packed);
// Second input of a split convolution (a ResNet block's projection shortcut folded into the block's second convolution): a channel-quad tensor of C
// channels on an H x W grid, read at (oh * sh, ow * sw) through a 1x1 filter whose C / 32 weight chunks (conv2d_split6_pack of the 1x1 layer) follow
// the main filter's in `packed`; X == nullptr: none.  Only where conv2d_split6_takes_second_input(g).
struct SecondInput {
  const float *X = nullptr;
  int C = 0, H = 0, W = 0, sh = 1, sw = 1;
};
bool conv2d_split6_takes_second_input(const ConvGeom &g);
void conv2d_split6(hipStream_t s, const float *X, const float *packed, const float *bias, const float *residual, float *Y, int64_t rows,
                   const ConvGeom &g, ActParam act, const SecondInput &x2 = SecondInput());
void pool2d(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int H, int W, int OH, int OW, int kh, int kw,
            int sh, int sw, int pt, int pl, int dh, int dw, bool is_max, bool count_pad, bool cq);
// y[n,c,p] = x[n,c,p] / (bias + alpha/size * sum_{c' in window(c)} x[n,c',p]^2)^beta over [rows, C, S] (cq: channel-quad planes)
void lrn(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int S, int size, float alpha, float beta, float bias, bool cq);
// Y[n, j*g + i, p] = X[n, i*(C/g) + j, p]
void channel_shuffle(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int S, int groups, bool cq);
void global_avgpool(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int S, bool cq, bool is_max = false);

}  // namespace infera_hip::kern
