// eltwise.hip -- HBM-bound elementwise kernels, row softmax and the synthetic-table fill.
//
// All of these stream each byte once: grid-stride loops, 16 B per lane where the layout allows,
// grid capped at 256 CUs x 8 blocks (cdna_hip_programming.md Guideline 11).
#include "device_common.hpp"

#include <cstdlib>

namespace infera_hip::kern {

namespace {

constexpr int kBlock = 256;
inline int grid_for(int64_t work_items) {
  int64_t g = (work_items + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return int(g);
}

__global__ __launch_bounds__(kBlock) void unary_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n,
                                                      ActParam act) {
  const int64_t n4 = n >> 2;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
  f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
    f32x4 v = x4[i];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = apply_act(v[j], act);
    y4[i] = v;
  }
  for (int64_t i = (n4 << 2) + int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) y[i] = apply_act(x[i], act);
}

__global__ __launch_bounds__(kBlock) void binary_const_kernel(const float *__restrict__ x, const float *__restrict__ c,
                                                             float *__restrict__ y, int64_t n, int64_t per_row, char op,
                                                             bool const_left, ActParam act) {
  // the position inside the row advances by (stride mod per_row) per trip: one 64-bit modulo per thread, not per element
  const int64_t stride = int64_t(gridDim.x) * kBlock, i0 = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t step = stride % per_row;
  int64_t j = i0 % per_row;
  for (int64_t i = i0; i < n; i += stride) {
    y[i] = apply_act(apply_bop(x[i], c[j], op, const_left), act);
    j += step;
    if (j >= per_row) j -= per_row;
  }
}

// Per-feature x*scale + shift on a [rows, C] table (S == 1): as above without the per-element division, and in
// 16-byte pieces when the rows are whole quads.
__global__ __launch_bounds__(kBlock) void affine_rows_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, float *__restrict__ y, int64_t n,
                                                            int64_t C, ActParam act, bool vec4) {
  const int64_t stride = int64_t(gridDim.x) * kBlock, i0 = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (vec4) {
    const int64_t n4 = n >> 2, C4 = C >> 2, step = stride % C4;
    int64_t j = i0 % C4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x), *sc4 = reinterpret_cast<const f32x4 *>(scale),
                *sh4 = reinterpret_cast<const f32x4 *>(shift);
    f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
    for (int64_t i = i0; i < n4; i += stride) {
      const f32x4 u = x4[i], a = sc4[j], b = sh4[j];
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; e++) r[e] = apply_act(u[e] * a[e] + b[e], act);
      y4[i] = r;
      j += step;
      if (j >= C4) j -= C4;
    }
    return;
  }
  const int64_t step = stride % C;
  int64_t j = i0 % C;
  for (int64_t i = i0; i < n; i += stride) {
    y[i] = apply_act(x[i] * scale[j] + shift[j], act);
    j += step;
    if (j >= C) j -= C;
  }
}

__global__ __launch_bounds__(kBlock) void binary_act_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                           float *__restrict__ y, int64_t n, char op, ActParam act) {
  const int64_t n4 = n >> 2;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const f32x4 *a4 = reinterpret_cast<const f32x4 *>(a);
  const f32x4 *b4 = reinterpret_cast<const f32x4 *>(b);
  f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
    f32x4 u = a4[i], v = b4[i], r;
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = apply_act(apply_bop(u[j], v[j], op, false), act);
    y4[i] = r;
  }
  for (int64_t i = (n4 << 2) + int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    y[i] = apply_act(apply_bop(a[i], b[i], op, false), act);
}

// Per-channel work on [N,C,S] tensors with a (plane, position block) grid: the plane -- n*C + c, or n*C/4 + c/4 in
// channel-quad planes -- is the workgroup's, so its scale / shift / gate values are scalar loads and no thread divides a
// flat 64-bit index (that decomposition was ~100 VALU instructions per element on kernels that move 8 bytes per element).
template <bool CQ>
__global__ __launch_bounds__(kBlock) void affine_planes_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                              const float *__restrict__ shift, float *__restrict__ y, int C, int S,
                                                              ActParam act) {
  const int pos = int(blockIdx.y) * kBlock + int(threadIdx.x);
  if (pos >= S) return;
  const int64_t plane = blockIdx.x;
  if constexpr (CQ) {
    const int c4 = int(plane % (C >> 2));
    const f32x4 sc = reinterpret_cast<const f32x4 *>(scale)[c4], sh = reinterpret_cast<const f32x4 *>(shift)[c4];
    const f32x4 u = reinterpret_cast<const f32x4 *>(x)[plane * S + pos];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = apply_act(u[e] * sc[e] + sh[e], act);
    reinterpret_cast<f32x4 *>(y)[plane * S + pos] = r;
  } else {
    const int c = int(plane % C);
    y[plane * S + pos] = apply_act(x[plane * S + pos] * scale[c] + shift[c], act);
  }
}

template <bool CQ>
__global__ __launch_bounds__(kBlock) void gate_planes_kernel(const float *__restrict__ a, const float *__restrict__ gate,
                                                            float *__restrict__ y, int C, int S, char op, ActParam act) {
  const int pos = int(blockIdx.y) * kBlock + int(threadIdx.x);
  if (pos >= S) return;
  const int64_t plane = blockIdx.x;  // the gate tensor [N,C] is exactly one value per NCHW plane / one quad per CQ plane
  if constexpr (CQ) {
    const f32x4 g = reinterpret_cast<const f32x4 *>(gate)[plane];
    const f32x4 u = reinterpret_cast<const f32x4 *>(a)[plane * S + pos];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = apply_act(apply_bop(u[e], g[e], op, false), act);
    reinterpret_cast<f32x4 *>(y)[plane * S + pos] = r;
  } else {
    y[plane * S + pos] = apply_act(apply_bop(a[plane * S + pos], gate[plane], op, false), act);
  }
}

// y[r, c, s] = act(a[r, c, s] (op) gate[r, c])   (squeeze-and-excitation style per-channel gate)
__global__ __launch_bounds__(kBlock) void binary_gate_kernel(const float *__restrict__ a, const float *__restrict__ gate,
                                                            float *__restrict__ y, int64_t n, int64_t C, int64_t S, char op, ActParam act,
                                                            bool cq) {
  const int64_t stride = int64_t(gridDim.x) * kBlock, per_row = C * S;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t row = i / per_row, j = i - row * per_row;
    const int64_t c = cq ? ((j >> 2) / S) * 4 + (j & 3) : j / S;
    y[i] = apply_act(apply_bop(a[i], gate[row * C + c], op, false), act);
  }
}

__global__ __launch_bounds__(kBlock) void affine_channel_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                               const float *__restrict__ shift, float *__restrict__ y,
                                                               int64_t n, int64_t C, int64_t S, ActParam act, bool cq) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t c = cq ? ((i >> 2) / S) % (C >> 2) * 4 + (i & 3) : (i / S) % C;
    y[i] = apply_act(x[i] * scale[c] + shift[c], act);
  }
}

// One lane per softmax vector when the vector is short (the common case here: 10 classes); the
// vector's `len` elements are `inner` floats apart.  max / exp / sum / divide in the oracle's order.
__global__ __launch_bounds__(kBlock) void softmax_small_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                              int64_t nvec, int64_t len, int64_t inner, int mode) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const bool logsm = mode == 1;
  for (int64_t v = int64_t(blockIdx.x) * kBlock + threadIdx.x; v < nvec; v += stride) {
    const int64_t ou = v / inner, in = v % inner;
    const float *src = x + ou * len * inner + in;
    float *dst = y + ou * len * inner + in;
    if (mode >= 2) {  // Normalizer: max|x| / sum|x| / sqrt(sum x^2), sequential like the oracle
      float d = 0.f;
      for (int64_t j = 0; j < len; j++) {
        const float u = src[j * inner], a = fabsf(u);
        d = mode == 2 ? fmaxf(d, a) : mode == 3 ? d + a : fmaf(u, u, d);
      }
      if (mode == 4) d = sqrtf(d);
      d = fmaxf(d, 1e-30f);
      for (int64_t j = 0; j < len; j++) dst[j * inner] = src[j * inner] / d;
      continue;
    }
    float mx = -INFINITY;
    for (int64_t j = 0; j < len; j++) mx = fmaxf(mx, src[j * inner]);
    float sum = 0.f;
    for (int64_t j = 0; j < len; j++) sum += expf(src[j * inner] - mx);
    if (logsm) {
      const float ls = logf(sum);
      for (int64_t j = 0; j < len; j++) dst[j * inner] = (src[j * inner] - mx) - ls;
    } else {
      for (int64_t j = 0; j < len; j++) dst[j * inner] = expf(src[j * inner] - mx) / sum;
    }
  }
}

// One wave per vector for long contiguous vectors (inner == 1): lanes stride the vector, the
// reductions are wave shuffles over 64 lanes.  The sum is therefore tree-ordered, not sequential.
__global__ __launch_bounds__(kBlock) void softmax_wave_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                             int64_t nvec, int64_t len, int mode) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kBlock) >> 6;
  const bool logsm = mode == 1;
  for (int64_t v = wave; v < nvec; v += nwaves) {
    const float *src = x + v * len;
    float *dst = y + v * len;
    if (mode >= 2) {  // Normalizer
      float d = 0.f;
      for (int64_t j = lane; j < len; j += 64) {
        const float u = src[j], a = fabsf(u);
        d = mode == 2 ? fmaxf(d, a) : mode == 3 ? d + a : fmaf(u, u, d);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) d = mode == 2 ? fmaxf(d, __shfl_xor(d, o)) : d + __shfl_xor(d, o);
      if (mode == 4) d = sqrtf(d);
      d = fmaxf(d, 1e-30f);
      for (int64_t j = lane; j < len; j += 64) dst[j] = src[j] / d;
      continue;
    }
    float mx = -INFINITY;
    for (int64_t j = lane; j < len; j += 64) mx = fmaxf(mx, src[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int64_t j = lane; j < len; j += 64) sum += expf(src[j] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float ls = logf(sum);
    for (int64_t j = lane; j < len; j += 64) dst[j] = logsm ? (src[j] - mx) - ls : expf(src[j] - mx) / sum;
  }
}

// Contiguous vectors of 5..1024 elements (class scores, inner == 1): LPR lanes share a vector and hold it in registers
// (EPL elements each, element j of lane l is l + LPR*j: consecutive lanes read consecutive floats), so the vector is
// read once and written once; max / sum meet in a butterfly over the LPR lanes.  (One lane per vector -- the kernel
// above -- walks 100 floats of its own row while its 63 neighbours do the same 400 bytes away: 15x slower at len 100.)
template <int LPR, int EPL>
__global__ __launch_bounds__(kBlock) void softmax_rows_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t nvec, int len,
                                                             int mode) {
  constexpr int VPW = 64 / LPR;  // vectors per wave
  const int lane = threadIdx.x & 63, sub = lane % LPR, slot = lane / LPR;
  const int64_t wave = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6, nwaves = (int64_t(gridDim.x) * kBlock) >> 6;
  for (int64_t v0 = wave * VPW; v0 < nvec; v0 += nwaves * VPW) {
    const int64_t v = v0 + slot;
    const bool live = v < nvec;
    const float *src = x + (live ? v : nvec - 1) * len;
    float e[EPL];
#pragma unroll
    for (int j = 0; j < EPL; j++) {
      const int k = sub + LPR * j;
      e[j] = k < len ? src[k] : 0.f;
    }
    float red = mode >= 2 ? 0.f : -INFINITY;  // softmax: max; Normalizer: max|x| / sum|x| / sum x^2
#pragma unroll
    for (int j = 0; j < EPL; j++)
      if (sub + LPR * j < len) {
        const float u = e[j], a = fabsf(u);
        red = mode < 2 ? fmaxf(red, u) : mode == 2 ? fmaxf(red, a) : mode == 3 ? red + a : fmaf(u, u, red);
      }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) {
      const float other = __shfl_xor(red, o);
      red = (mode < 2 || mode == 2) ? fmaxf(red, other) : red + other;
    }
    if (mode >= 2) {
      if (mode == 4) red = sqrtf(red);
      red = fmaxf(red, 1e-30f);
#pragma unroll
      for (int j = 0; j < EPL; j++) e[j] = e[j] / red;
    } else {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < EPL; j++)
        if (sub + LPR * j < len) {
          const float t = expf(e[j] - red);
          sum += t;
          e[j] = mode == 0 ? t : e[j] - red;
        }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float ls = logf(sum);
#pragma unroll
      for (int j = 0; j < EPL; j++) e[j] = mode == 0 ? e[j] / sum : e[j] - ls;
    }
    if (live) {
      float *dst = y + v * len;
#pragma unroll
      for (int j = 0; j < EPL; j++) {
        const int k = sub + LPR * j;
        if (k < len) dst[k] = e[j];
      }
    }
  }
}

// Column-block copy between row-major matrices: dst[r, dst_off : dst_off+len] = src[r, src_off : src_off+len].
// One piece of a Concat along the feature / channel axis (src is a whole row) or a Slice / Split / one input of a
// multi-input model (dst is a whole row).  16-byte moves when every offset and stride is a multiple of 4 floats.
__global__ __launch_bounds__(kBlock) void copy_cols_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t rows,
                                                          int64_t len, int64_t src_stride, int64_t src_off, int64_t dst_stride,
                                                          int64_t dst_off, bool vec4) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  if (vec4) {
    const int64_t l4 = len >> 2, n4 = rows * l4;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
      const int64_t r = i / l4, c = i - r * l4;
      *reinterpret_cast<f32x4 *>(dst + r * dst_stride + dst_off + 4 * c) = *reinterpret_cast<const f32x4 *>(src + r * src_stride + src_off + 4 * c);
    }
    return;
  }
  const int64_t n = rows * len;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / len, c = i - r * len;
    dst[r * dst_stride + dst_off + c] = src[r * src_stride + src_off + c];
  }
}

// dst[r, 0:K] = src[r, 0:K], dst[r, K:Kp] = 0   (rows padded to a multiple of 4 floats for 16-byte operand loads)
__global__ __launch_bounds__(kBlock) void pad_cols_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t rows, int64_t K,
                                                         int64_t Kp) {
  const int64_t stride = int64_t(gridDim.x) * kBlock, n = rows * Kp;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / Kp, c = i - r * Kp;
    dst[i] = c < K ? src[r * K + c] : 0.f;
  }
}

// Column-major staging -> row-major table: src is [ncols][rows] (each DuckDB flat column copied as the contiguous
// run it already is), dst is [rows][ncols].  32x32 tiles through LDS (+1 padding: conflict-free both ways), both
// sides coalesced.  The host no longer transposes; this costs ~2 us per 2048 x 128 chunk on the GPU.
__global__ __launch_bounds__(kBlock) void transpose_cm_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t rows,
                                                             int64_t ncols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
  const int64_t r0 = int64_t(blockIdx.x) * 32, c0 = int64_t(blockIdx.y) * 32;
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int64_t c = c0 + j, r = r0 + tx;
    tile[j][tx] = (c < ncols && r < rows) ? src[c * rows + r] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int64_t r = r0 + j, c = c0 + tx;
    if (r < rows && c < ncols) dst[r * ncols + c] = tile[tx][j];
  }
}

// Index of the first maximum of each row, as an f32 value (NaN never wins: ONNX ArgMax on comparisons).
__global__ __launch_bounds__(kBlock) void argmax_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t rows, int64_t len) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < rows; r += stride) {
    const float *src = x + r * len;
    float best = src[0];
    int64_t bi = 0;
    for (int64_t j = 1; j < len; j++)
      if (src[j] > best) {
        best = src[j];
        bi = j;
      }
    y[r] = float(bi);
  }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(kBlock) void synth_fill_kernel(float *__restrict__ dst, uint64_t seed, uint64_t base,
                                                           uint64_t n) {
  const uint64_t stride = uint64_t(gridDim.x) * kBlock;
  for (uint64_t i = uint64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const uint64_t u = splitmix64(seed ^ (base + i));
    dst[i] = float(u >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
  }
}

}  // namespace

void unary(hipStream_t s, const float *x, float *y, int64_t n, ActParam act) {
  if (n <= 0) return;
  hipLaunchKernelGGL(unary_kernel, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, s, x, y, n, act);
}

void binary_const(hipStream_t s, const float *x, const float *c, float *y, int64_t rows, int64_t per_row, char op,
                  bool const_left, ActParam act) {
  const int64_t n = rows * per_row;
  if (n <= 0) return;
  hipLaunchKernelGGL(binary_const_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, x, c, y, n, per_row, op, const_left, act);
}

void binary_act(hipStream_t s, const float *a, const float *b, float *y, int64_t n, char op, ActParam act) {
  if (n <= 0) return;
  hipLaunchKernelGGL(binary_act_kernel, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, s, a, b, y, n, op, act);
}

void binary_gate(hipStream_t s, const float *a, const float *gate, float *y, int64_t rows, int64_t C, int64_t S, char op, ActParam act,
                 bool cq) {
  const int64_t n = rows * C * S;
  if (n <= 0) return;
  if (S <= 65535LL * kBlock && (!cq || C % 4 == 0)) {
    const dim3 grid(unsigned(cq ? rows * C / 4 : rows * C), unsigned((S + kBlock - 1) / kBlock));
    if (cq) hipLaunchKernelGGL(gate_planes_kernel<true>, grid, dim3(kBlock), 0, s, a, gate, y, int(C), int(S), op, act);
    else hipLaunchKernelGGL(gate_planes_kernel<false>, grid, dim3(kBlock), 0, s, a, gate, y, int(C), int(S), op, act);
    return;
  }
  hipLaunchKernelGGL(binary_gate_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, a, gate, y, n, C, S, op, act, cq);
}

void affine_channel(hipStream_t s, const float *x, const float *scale, const float *shift, float *y, int64_t rows,
                    int64_t C, int64_t S, ActParam act, bool cq) {
  const int64_t n = rows * C * S;
  if (n <= 0) return;
  if (S == 1) {  // a table: no layout to respect
    const bool vec4 = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    hipLaunchKernelGGL(affine_rows_kernel, dim3(grid_for(vec4 ? n / 4 : n)), dim3(kBlock), 0, s, x, scale, shift, y, n, C, act, vec4);
    return;
  }
  if (S <= 65535LL * kBlock && (!cq || C % 4 == 0)) {
    const dim3 grid(unsigned(cq ? rows * C / 4 : rows * C), unsigned((S + kBlock - 1) / kBlock));
    if (cq) hipLaunchKernelGGL(affine_planes_kernel<true>, grid, dim3(kBlock), 0, s, x, scale, shift, y, int(C), int(S), act);
    else hipLaunchKernelGGL(affine_planes_kernel<false>, grid, dim3(kBlock), 0, s, x, scale, shift, y, int(C), int(S), act);
    return;
  }
  hipLaunchKernelGGL(affine_channel_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, x, scale, shift, y, n, C, S, act, cq);
}

void softmax(hipStream_t s, const float *x, float *y, int64_t rows, int64_t outer, int64_t len, int64_t inner, int mode) {
  const int64_t nvec = rows * outer * inner;
  if (nvec <= 0 || len <= 0) return;
  static const bool rows_kernel = !(getenv("INFERA_SOFTMAX_ROWS") && atoi(getenv("INFERA_SOFTMAX_ROWS")) == 0);
  if (rows_kernel && inner == 1 && len >= 5 && len <= 1024) {
    auto go = [&](auto kernel, int lpr) {
      hipLaunchKernelGGL(kernel, dim3(grid_for((nvec + 64 / lpr - 1) / (64 / lpr) * 64)), dim3(kBlock), 0, s, x, y, nvec, int(len), mode);
    };
    if (len <= 16) go(softmax_rows_kernel<16, 1>, 16);
    else if (len <= 64) go(softmax_rows_kernel<16, 4>, 16);
    else if (len <= 128) go(softmax_rows_kernel<32, 4>, 32);
    else if (len <= 256) go(softmax_rows_kernel<64, 4>, 64);
    else go(softmax_rows_kernel<64, 16>, 64);
  } else if (inner == 1 && len >= 256) {
    hipLaunchKernelGGL(softmax_wave_kernel, dim3(grid_for(nvec * 64)), dim3(kBlock), 0, s, x, y, nvec, len, mode);
  } else {
    // vectors index as (row*outer + ou, in): flatten (row,outer) into the `ou` coordinate
    hipLaunchKernelGGL(softmax_small_kernel, dim3(grid_for(nvec)), dim3(kBlock), 0, s, x, y, nvec, len, inner, mode);
  }
}

void copy_cols(hipStream_t s, const float *src, float *dst, int64_t rows, int64_t len, int64_t src_stride, int64_t src_off,
               int64_t dst_stride, int64_t dst_off) {
  if (rows <= 0 || len <= 0) return;
  const bool vec4 = ((len | src_stride | src_off | dst_stride | dst_off) & 3) == 0;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(vec4 ? rows * len / 4 : rows * len)), dim3(kBlock), 0, s, src, dst, rows, len,
                     src_stride, src_off, dst_stride, dst_off, vec4);
}

void pad_cols(hipStream_t s, const float *src, float *dst, int64_t rows, int64_t K, int64_t Kp) {
  if (rows <= 0 || Kp <= 0) return;
  hipLaunchKernelGGL(pad_cols_kernel, dim3(grid_for(rows * Kp)), dim3(kBlock), 0, s, src, dst, rows, K, Kp);
}

// ---- zero-copy column gather (round 3; launch shape round 5) ----------------------------------------------------------------------
// The caller's column runs live in host memory that the application REGISTERED with the runtime (infera_hip_register_host_memory):
// the GPU reads them in place over PCIe and writes the column-major f32 chunk [ncols][rows] into HBM -- the CPU never touches the
// data.  ONE WAVE per (column, 2048-row block): a wave instruction reads 1 KB of one column run (16 B per lane, contiguous) and all
// eight of a FLOAT run's are issued before the first store -- the whole 8 KB run of a DataChunk column is in flight from one wave.
// Casts as the reference's ExtractFeatures (static_cast<float>, infera_extension.cpp:211-222): f64 -> f32 and i64 -> f32 round to
// nearest even (v_cvt_f32_f64; __ll2float_rn), i32 -> f32 exact rounding; a constant vector broadcasts its one value.  Runs whose
// address is not 16-byte aligned are read element-wise.
// Launch shape, measured with nothing else in the queues (tools/ubench/chunk_pull_probe.hip, 1 MiB fetches back to back on 1 / 4
// streams): this shape 49.7 / 54.3 GB/s; round 3's one 256-lane workgroup per column with two loads per lane 48.1 / 51.0; one-wave
// workgroups with two loads 43.1 / 47.5; hipMemcpy2DAsync 36.7 / 36.7 -- a 2-D copy runs ALONE however many streams issue them.
constexpr int kGatherRowBlock = 2048;
template <class T, class Cvt>
__device__ __forceinline__ void gather_run(const T *__restrict__ s, float *__restrict__ d, int n, int lane, Cvt cvt) {
  for (int base = 0; base < n; base += 64 * 8) {
    T v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int idx = base + lane + 64 * i;
      if (idx < n) v[i] = s[idx];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int idx = base + lane + 64 * i;
      if (idx < n) d[idx] = cvt(v[i]);
    }
  }
}
__device__ __forceinline__ void gather_cols_wave(const ColumnTable &tab, int ncols, int64_t rows, float *__restrict__ dst) {
  const int c = int(blockIdx.x) * (kBlock / 64) + int(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= ncols) return;
  const int type = tab.type[c] & 7;
  const bool constant = tab.type[c] & 8;
  const char *src = static_cast<const char *>(tab.ptr[c]);
  const int64_t r0 = int64_t(blockIdx.y) * kGatherRowBlock;
  const int n = int(min<int64_t>(rows - r0, kGatherRowBlock));
  float *d = dst + int64_t(c) * rows + r0;
  if (constant) {
    float v;
    if (type == 0) v = *reinterpret_cast<const float *>(src);
    else if (type == 1) v = float(*reinterpret_cast<const double *>(src));
    else if (type == 2) v = float(*reinterpret_cast<const int *>(src));
    else v = __ll2float_rn(*reinterpret_cast<const long long *>(src));
    for (int r = lane; r < n; r += 64) d[r] = v;
    return;
  }
  if (type == 0) {
    const float *s = reinterpret_cast<const float *>(src) + r0;
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0 && (n & 3) == 0) {
      const f32x4 *s4 = reinterpret_cast<const f32x4 *>(s);
      f32x4 *d4 = reinterpret_cast<f32x4 *>(d);
      const int n4 = n >> 2;  // (<= 512: one round of eight loads per lane)
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (lane + 64 * i < n4) v[i] = s4[lane + 64 * i];
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (lane + 64 * i < n4) d4[lane + 64 * i] = v[i];
      return;
    }
    if ((reinterpret_cast<uintptr_t>(s) & 15) == 8 && (reinterpret_cast<uintptr_t>(d) & 15) == 0 && (n & 3) == 0) {
      // A run that starts 8 bytes past a 16-byte boundary: what a FLAT vector inside a DuckDB buffer-manager block is (the block header is 8
      // bytes; round 6).  Read as the n/4 + 1 ALIGNED 16-byte pieces that cover it -- the two floats in front of the run and the two behind it
      // share their 16-byte unit, hence their page, with the run's own first / last floats: no access leaves the pages the run lies in -- and
      // shifted by half a piece on the way out (the upper half of piece j + the lower half of piece j + 1, taken from the neighbouring lane).
      // Same 1 KB wave instructions, all in flight before the first store, as the aligned path; element-wise 4-byte loads were 3x slower.
      const f32x4 *a4 = reinterpret_cast<const f32x4 *>(s - 2);
      f32x4 *d4 = reinterpret_cast<f32x4 *>(d);
      const int n4 = n >> 2;  // (<= 512)
      f32x4 v[9];
#pragma unroll
      for (int i = 0; i < 9; i++) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (lane + 64 * i <= n4) v[i] = a4[lane + 64 * i];
      if (lane == 0 && n4 == 512) v[8] = a4[512];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float nx = __shfl_down(v[i][0], 1), ny = __shfl_down(v[i][1], 1);
        const float fx = __shfl(v[i + 1][0], 0), fy = __shfl(v[i + 1][1], 0);  // lane 63's neighbour: lane 0 of the next round
        if (lane == 63) nx = fx, ny = fy;
        if (lane + 64 * i < n4) d4[lane + 64 * i] = f32x4{v[i][2], v[i][3], nx, ny};
      }
      return;
    }
    gather_run(s, d, n, lane, [](float x) { return x; });
  } else if (type == 1) {
    gather_run(reinterpret_cast<const double *>(src) + r0, d, n, lane, [](double x) { return float(x); });
  } else if (type == 2) {
    gather_run(reinterpret_cast<const int *>(src) + r0, d, n, lane, [](int x) { return float(x); });
  } else {
    gather_run(reinterpret_cast<const long long *>(src) + r0, d, n, lane, [](long long x) { return __ll2float_rn(x); });
  }
}

__global__ __launch_bounds__(kBlock) void gather_cols_kernel(ColumnTable tab, int ncols, int64_t rows, float *__restrict__ dst) {
  gather_cols_wave(tab, ncols, rows, dst);
}

void gather_columns_device(hipStream_t s, const ColumnTable &tab, int ncols, int64_t rows, float *dst) {
  if (rows <= 0 || ncols <= 0) return;
  // (Also measured and dropped: the fused MLP's tile kernel reading the column runs ITSELF (each workgroup pulling its own 32 rows of
  // every column into LDS: one launch per chunk, bit-identical) -- 46 M rows/s against 82: 128-byte pieces per (column, tile) use the
  // link far worse than this kernel's 1 KB wave instructions; more hardware queues (GPU_MAX_HW_QUEUES=8 / 16): worse, fewer (1 / 2 / 3): worse;
  // round 6: pulls of different callers taking TURNS on the link -- a per-GPU ticket in device memory, a pull's waves asleep until fewer than
  // K = 1 / 2 pulls that started before it were unfinished (the idea: four callers' pulls that start together share the link, finish together
  // and idle it together) -- bit-identical, within +-3 % at every caller count on DuckDB-shaped segments: profiles/r06_duckdb_blocks.txt.)
  hipLaunchKernelGGL(gather_cols_kernel, dim3(unsigned((ncols + kBlock / 64 - 1) / (kBlock / 64)), unsigned((rows + kGatherRowBlock - 1) / kGatherRowBlock)),
                     dim3(kBlock), 0, s, tab, ncols, rows, dst);
}

void transpose_cm(hipStream_t s, const float *src, float *dst, int64_t rows, int64_t ncols) {
  if (rows <= 0 || ncols <= 0) return;
  hipLaunchKernelGGL(transpose_cm_kernel, dim3(unsigned((rows + 31) / 32), unsigned((ncols + 31) / 32)), dim3(kBlock), 0, s, src, dst, rows, ncols);
}

// Across-channel LRN, one lane per element.  Element (n, c, p) sits at ((n*C + c)*S + p) in NCHW and at
// ((n*C/4 + c/4)*S + p)*4 + c%4 in channel-quad planes; the window walks channels at a fixed position p.
__global__ __launch_bounds__(kBlock) void lrn_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n, int C, int S,
                                                    int size, float alpha_over_size, float beta, float bias, bool cq) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const int lo = (size - 1) / 2, hi = size - 1 - lo;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    int64_t img, p;
    int c;
    if (cq) {
      const int64_t quad = i >> 2, plane = quad / S;  // plane = img * (C/4) + c/4
      p = quad - plane * S;
      img = plane / (C >> 2);
      c = int(plane - img * (C >> 2)) * 4 + int(i & 3);
    } else {
      const int64_t nc = i / S;
      p = i - nc * S;
      img = nc / C;
      c = int(nc - img * C);
    }
    const int c0 = max(c - lo, 0), c1 = min(c + hi, C - 1);
    float sq = 0.f;
    for (int k = c0; k <= c1; k++) {
      const float v = cq ? x[((img * (C >> 2) + (k >> 2)) * S + p) * 4 + (k & 3)] : x[(img * C + k) * S + p];
      sq = fmaf(v, v, sq);
    }
    y[i] = x[i] / powf(bias + alpha_over_size * sq, beta);
  }
}

// Channel shuffle, one lane per output element: out channel j*g + i <- in channel i*(C/g) + j
__global__ __launch_bounds__(kBlock) void channel_shuffle_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n, int C,
                                                                int S, int g, bool cq) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const int per = C / g;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    int64_t img, p;
    int c;
    if (cq) {
      const int64_t quad = i >> 2, plane = quad / S;
      p = quad - plane * S;
      img = plane / (C >> 2);
      c = int(plane - img * (C >> 2)) * 4 + int(i & 3);
    } else {
      const int64_t nc = i / S;
      p = i - nc * S;
      img = nc / C;
      c = int(nc - img * C);
    }
    const int k = (c % g) * per + c / g;
    y[i] = cq ? x[((img * (C >> 2) + (k >> 2)) * S + p) * 4 + (k & 3)] : x[(img * C + k) * S + p];
  }
}

void lrn(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int S, int size, float alpha, float beta, float bias, bool cq) {
  const int64_t n = rows * C * S;
  if (n <= 0) return;
  hipLaunchKernelGGL(lrn_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, X, Y, n, C, S, size, alpha / float(size), beta, bias, cq);
}

// (plane, position block) grid as for the other per-channel kernels: the four source channels of an output quad are
// scalar; NCHW planes are plain contiguous copies.
template <bool CQ>
__global__ __launch_bounds__(kBlock) void channel_shuffle_planes_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int S, int g) {
  const int pos = int(blockIdx.y) * kBlock + int(threadIdx.x);
  if (pos >= S) return;
  const int64_t plane = blockIdx.x;
  const int per = C / g;
  if constexpr (CQ) {
    const int C4 = C >> 2, cq = int(plane % C4);
    const int64_t img = plane / C4;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int c = 4 * cq + e, k = (c % g) * per + c / g;
      r[e] = x[((img * C4 + (k >> 2)) * S + pos) * 4 + (k & 3)];
    }
    reinterpret_cast<f32x4 *>(y)[plane * S + pos] = r;
  } else {
    const int c = int(plane % C), k = (c % g) * per + c / g;
    y[plane * S + pos] = x[(plane - c + k) * S + pos];
  }
}

void channel_shuffle(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int S, int groups, bool cq) {
  const int64_t n = rows * C * S;
  if (n <= 0) return;
  if (S <= 65535LL * kBlock && (!cq || C % 4 == 0)) {
    const dim3 grid(unsigned(cq ? rows * C / 4 : rows * C), unsigned((S + kBlock - 1) / kBlock));
    if (cq) hipLaunchKernelGGL(channel_shuffle_planes_kernel<true>, grid, dim3(kBlock), 0, s, X, Y, C, S, groups);
    else hipLaunchKernelGGL(channel_shuffle_planes_kernel<false>, grid, dim3(kBlock), 0, s, X, Y, C, S, groups);
    return;
  }
  hipLaunchKernelGGL(channel_shuffle_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, X, Y, n, C, S, groups, cq);
}

// Longer rows: LPR lanes share a row (consecutive lanes read consecutive floats), each keeps the first maximum of its
// elements, then a butterfly over the LPR lanes picks the winner (ties -> lowest index, like the sequential scan).
template <int LPR>
__global__ __launch_bounds__(kBlock) void argmax_rows_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t rows, int len) {
  constexpr int VPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR, slot = lane / LPR;
  const int64_t wave = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6, nwaves = (int64_t(gridDim.x) * kBlock) >> 6;
  for (int64_t r0 = wave * VPW; r0 < rows; r0 += nwaves * VPW) {
    const int64_t r = r0 + slot;
    const float *src = x + (r < rows ? r : rows - 1) * len;
    float best = -INFINITY;
    int bi = len;  // lanes without an element never win a tie
    for (int k = sub; k < len; k += LPR) {
      const float u = src[k];
      if (k == 0 || u > best) {  // element 0 starts the scan whatever it is (NaN included), like the sequential kernel
        best = u;
        bi = k;
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (oi < len && (bi == len || ov > best || (ov == best && oi < bi))) {
        best = ov;
        bi = oi;
      }
    }
    if (r < rows && sub == 0) y[r] = float(bi);
  }
}

void argmax_rows(hipStream_t s, const float *x, float *y, int64_t rows, int64_t len) {
  if (rows <= 0 || len <= 0) return;
  if (len >= 8 && len <= (1 << 30)) {
    auto go = [&](auto kernel, int lpr) {
      hipLaunchKernelGGL(kernel, dim3(grid_for((rows + 64 / lpr - 1) / (64 / lpr) * 64)), dim3(kBlock), 0, s, x, y, rows, int(len));
    };
    if (len <= 64) go(argmax_rows_kernel<16>, 16);
    else if (len <= 256) go(argmax_rows_kernel<32>, 32);
    else go(argmax_rows_kernel<64>, 64);
    return;
  }
  hipLaunchKernelGGL(argmax_kernel, dim3(grid_for(rows)), dim3(kBlock), 0, s, x, y, rows, len);
}

void synth_fill(hipStream_t s, float *dst, uint64_t seed, uint64_t row0, uint64_t rows, uint64_t ncols) {
  const uint64_t n = rows * ncols;
  if (n == 0) return;
  hipLaunchKernelGGL(synth_fill_kernel, dim3(grid_for(int64_t(n))), dim3(kBlock), 0, s, dst, seed, row0 * ncols, n);
}

}  // namespace infera_hip::kern
