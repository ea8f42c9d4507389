// plan.hpp -- the lowered, device-independent execution plan of one model.
//
// infera_load_model turns the ONNX graph into this at load time (the reference does the analogous
// work with Tract's into_optimized()/into_runnable(), engine.rs:52-55).  Every activation is a
// row-major [rows, per_row] f32 matrix whose leading axis is the table-row (batch) axis, so all
// steps are row-independent and a scan shards by row range with no exchange.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "onnx_model.hpp"

namespace infera_hip {

// Kinds 1..5 may be fused into the epilogue of a Dense / Conv2d step (the MFMA kernels resolve them at
// compile time); the rest run in the elementwise kernels (fused into Binary*/AffineChannel or as a Unary step).
enum class Act : int {
  None = 0, Relu = 1, Sigmoid = 2, Tanh = 3, LeakyRelu = 4, Clip = 5,
  Exp = 6, Log = 7, Sqrt = 8, Neg = 9, Abs = 10, Elu = 11, Selu = 12, Softplus = 13, HardSigmoid = 14, HardSwish = 15,
  Erf = 16, Gelu = 17, Reciprocal = 18, Floor = 19, Ceil = 20, Softsign = 21, Trunc = 22, Round = 23,
  Swish = 24,  // x * sigmoid(x): recognised from Mul(x, Sigmoid(x)) (EfficientNet / YOLO exports)
};
constexpr int kMaxMfmaFusedAct = 5;  // the load-time specialised chain kernels resolve kinds 0..5
// what the ahead-of-time MFMA epilogues resolve (device_common.hpp dispatch_act): the above + the mobile-net gates
inline bool mfma_fusable(Act a) { return int(a) <= kMaxMfmaFusedAct || a == Act::HardSigmoid || a == Act::HardSwish || a == Act::Swish; }

enum class StepKind : int {
  Dense = 0,        // Y[rows,M] = act(X[rows,K] . W[K,M] + bias[M])       (MatMul / Gemm [+Add] [+act])
  Unary = 1,        // elementwise activation
  AffineChannel = 2,// y = x*scale[c] + shift[c]  per channel              (unfused BatchNormalization)
  BinaryConst = 3,  // y = x (op) cst[per_row]  (constant pre-broadcast to one row)
  BinaryAct = 4,    // y = a (op) b             (residual adds; S > 1: b is a per-channel gate [rows, C] broadcast over S positions)
  Softmax = 5,      // softmax / log-softmax over `sm_len` with (outer, len, inner) strides inside a row
  Conv2d = 6,       // NCHW convolution as implicit GEMM (BatchNormalization folded when adjacent)
  Pool2d = 7,       // MaxPool / AveragePool
  GlobalAvgPool = 8,
  CopyCols = 9,     // out[r, col_off : col_off+len] = in0[r, :]   (one piece of a Concat along the feature/channel axis)
  ArgMax = 10,      // out[r, 0] = float(index of the first maximum of in0[r, 0:len])   (labels as f32 values)
  SliceCols = 11,   // out[r, :] = in0[r, col_off : col_off+K]   (Slice / Split on the feature axis; one input of a multi-input model)
  LRN = 13,         // across-channel local response normalisation: y = x / (act_b' ... see lrn_* fields) over [N,C,S]
  ChannelShuffle = 14,  // out[n, j*g + i, p] = in0[n, i*(C/g) + j, p]   (Reshape [N,g,C/g,..] -> Transpose(0,2,1,..) -> Reshape; groups in `groups`)
  PadCols = 12,     // out[r, 0:K] = in0[r, :], zeros up to M columns   (row length -> multiple of 4 for the 16-byte loads of the MFMA kernels)
};

struct Step {
  StepKind kind = StepKind::Unary;
  int in0 = -1, in1 = -1, out = -1;  // activation buffer ids; 0 is the model input
  Act act = Act::None;               // fused trailing activation
  float act_a = 0.f, act_b = 0.f;    // LeakyRelu alpha / Clip lo,hi
  // Dense
  int64_t K = 0, M = 0;
  std::vector<float> W;     // [K, M] row-major (Gemm transB / alpha already folded)
  std::vector<float> bias;  // [M] or empty (Gemm beta folded)
  // AffineChannel: scale/shift per channel, S = elements per channel
  std::vector<float> scale, shift;
  int64_t S = 1;
  // BinaryConst / BinaryAct: + - * /  m(in) M(ax) ^(pow)  p(relu: x >= 0 ? x : c*x, BinaryConst only)
  char bop = '+';
  bool const_left = false;
  std::vector<float> cst;  // per_row elements
  // CopyCols: destination column offset (elements inside a row), length = in0's per_row.  SliceCols: SOURCE offset, length K
  int64_t col_off = 0;
  // Softmax
  int64_t sm_outer = 1, sm_len = 1, sm_inner = 1;
  bool log_softmax = false;
  int sm_norm = 0;  // 0: softmax family; row normalisation instead (ai.onnx.ml Normalizer): 1 MAX, 2 L1, 3 L2
  // Conv2d / Pool2d / GlobalAvgPool geometry (per sample)
  int64_t C = 0, H = 0, Wd = 0, Mo = 0, OH = 0, OW = 0;
  int64_t kh = 1, kw = 1, sh = 1, sw = 1, pt = 0, pl = 0, pb = 0, pr = 0, dh = 1, dw = 1, groups = 1;
  bool is_max = false, count_pad = false;
  // LRN: window `lrn_size` channels, y = x / (lrn_bias + lrn_alpha / lrn_size * sum x^2)^lrn_beta
  int64_t lrn_size = 0;
  float lrn_alpha = 1e-4f, lrn_beta = 0.75f, lrn_bias = 1.f;
  std::string origin;  // ONNX node names/ops this step came from (diagnostics)
};

struct Plan {
  std::vector<int64_t> input_shape, output_shape;  // -1 = symbolic (engine.rs:64-73)
  std::vector<std::vector<int64_t>> buf_shape;     // per activation buffer, dim0 = -1 or the fixed batch
  std::vector<int64_t> buf_per_row;                // elements per row of each buffer
  std::vector<Step> steps;
  int out_buf = 0;
  int64_t fixed_batch = -1;  // > 0 when the model's leading dim is a constant (e.g. linear.onnx [1,3])
  int64_t opset = 1;
  std::string output_name;         // the served graph output
  std::string output_declared_type;  // "" for f32; "int64" / "int32" when the graph declares an integer output that is served as f32 VALUES

  int64_t in_per_row() const { return buf_per_row[0]; }
  int64_t out_per_row() const { return buf_per_row[out_buf]; }
  double flops_per_row() const;  // 2*MACs of Dense/Conv steps (bias/activation excluded)
  std::string describe_json() const;
};

// Throws InferaError::onnx on unsupported graphs.
// output_select: "" = the first graph output (the reference's behaviour), else an output's name or decimal index.
Plan lower_model(const onnx::Model &m, const std::string &output_select = "");

}  // namespace infera_hip
