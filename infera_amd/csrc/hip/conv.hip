// conv.hip -- NCHW convolution / pooling kernels for the BLOB (image) path (BASELINE config C5).
//
// conv2d is an implicit GEMM on the exact-fp32 matrix cores, in the same transposed formulation as
// dense.hip:   Out^T[m, p] = sum_k  Wt[m, k] * col[k, p],   k = (c, kh, kw),  p = (n, oh, ow)
//   A operand = weights  (lane: m = lane&31, k = lane>>5)      -- read straight from L1/L2
//   B operand = im2col gather of the input (lane: p = lane&31, k = lane>>5) -- computed on the fly,
//               never materialised in HBM.
// A wave owns 32 output pixels x (MT*32) output channels.  Bias (with BatchNormalization already
// folded in by the loader), an optional residual add and the activation are fused in the epilogue.
#include "device_common.hpp"

namespace infera_hip::kern {

namespace {

constexpr int kBlock = 256;

template <int MT>
__global__ __launch_bounds__(kBlock) void conv2d_kernel(const float *__restrict__ X, const float *__restrict__ Wt,
                                                       const float *__restrict__ bias, const float *__restrict__ residual,
                                                       float *__restrict__ Y, int64_t total_pix, ConvGeom g, ActParam act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int Cg = g.C / g.groups, Mg = g.M / g.groups;
  const int KK = Cg * g.kh * g.kw;
  const int OHW = g.OH * g.OW;
  const int mtiles = (Mg + 32 * MT - 1) / (32 * MT);
  // blockIdx.x -> pixel tile (4 waves x 32 pixels), blockIdx.y -> (group, m-tile)
  const int grp = blockIdx.y / mtiles, mt0 = (blockIdx.y % mtiles) * 32 * MT;
  const int64_t pix = (int64_t(blockIdx.x) * 4 + wave) * 32 + r;
  const bool pvalid = pix < total_pix;
  const int64_t n = pvalid ? pix / OHW : 0;
  const int prem = pvalid ? int(pix % OHW) : 0;
  const int oh = prem / g.OW, ow = prem % g.OW;
  const int ih0 = oh * g.sh - g.pt, iw0 = ow * g.sw - g.pl;
  const float *xin = X + (n * g.C + int64_t(grp) * Cg) * g.H * g.W;
  const float *wg = Wt + int64_t(grp) * Mg * KK;

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

  const int khw = g.kh * g.kw;
  for (int k0 = 0; k0 < KK; k0 += 2) {
    const int k = k0 + h;
    float b = 0.f;
    if (pvalid && k < KK) {
      const int c = k / khw, rem = k % khw, ky = rem / g.kw, kx = rem % g.kw;
      const int iy = ih0 + ky * g.dh, ix = iw0 + kx * g.dw;
      if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) b = xin[(int64_t(c) * g.H + iy) * g.W + ix];
    }
#pragma unroll
    for (int t = 0; t < MT; t++) {
      const int m = mt0 + 32 * t + r;
      const float a = (m < Mg && k < KK) ? wg[int64_t(m) * KK + k] : 0.f;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  }

  if (!pvalid) return;
  // lane (r,h) holds pixel `pix`, channels mt0 + 32t + 8*(i>>2) + 4h + (i&3)
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int ml = mt0 + 32 * t + 8 * (i >> 2) + 4 * h + (i & 3);
      if (ml < Mg) {
        const int m = grp * Mg + ml;
        const int64_t o = (n * g.M + m) * OHW + prem;
        float v = acc[t][i];
        if (bias) v += bias[m];
        if (residual) v += residual[o];
        Y[o] = apply_act(v, act);
      }
    }
}

__global__ __launch_bounds__(kBlock) void pool2d_kernel(const float *__restrict__ X, float *__restrict__ Y, int64_t total,
                                                       int H, int W, int OH, int OW, int kh, int kw, int sh, int sw, int pt,
                                                       int pl, int dh, int dw, bool is_max, bool count_pad) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t o = int64_t(blockIdx.x) * kBlock + threadIdx.x; o < total; o += stride) {
    const int ow = int(o % OW), oh = int((o / OW) % OH);
    const int64_t nc = o / (int64_t(OW) * OH);
    const float *src = X + nc * H * W;
    float acc = is_max ? -INFINITY : 0.f;
    int cnt = 0;
    for (int i = 0; i < kh; i++)
      for (int j = 0; j < kw; j++) {
        const int iy = oh * sh - pt + i * dh, ix = ow * sw - pl + j * dw;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float v = src[iy * W + ix];
        acc = is_max ? fmaxf(acc, v) : acc + v;
        cnt++;
      }
    if (!is_max) acc = acc / float(count_pad ? kh * kw : (cnt ? cnt : 1));
    Y[o] = acc;
  }
}

// One wave per (n, c): sequential partial sums per lane, then a 64-lane shuffle tree.
__global__ __launch_bounds__(kBlock) void global_avgpool_kernel(const float *__restrict__ X, float *__restrict__ Y,
                                                               int64_t nc_total, int S) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kBlock) >> 6;
  for (int64_t nc = wave; nc < nc_total; nc += nwaves) {
    const float *src = X + nc * S;
    float acc = 0.f;
    for (int i = lane; i < S; i += 64) acc += src[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) Y[nc] = acc / float(S);
  }
}

inline int grid_for(int64_t items) {
  int64_t g = (items + kBlock - 1) / kBlock;
  return int(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

void conv2d(hipStream_t s, const float *X, const float *Wt, const float *bias, const float *residual, float *Y,
            int64_t rows, const ConvGeom &g, ActParam act) {
  const int64_t total_pix = rows * g.OH * g.OW;
  if (total_pix <= 0) return;
  const int Mg = g.M / g.groups;
  const unsigned bx = unsigned((total_pix + 127) / 128);
  if (Mg <= 32) {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 31) / 32)));
    hipLaunchKernelGGL(conv2d_kernel<1>, grid, dim3(kBlock), 0, s, X, Wt, bias, residual, Y, total_pix, g, act);
  } else if (Mg <= 64) {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 63) / 64)));
    hipLaunchKernelGGL(conv2d_kernel<2>, grid, dim3(kBlock), 0, s, X, Wt, bias, residual, Y, total_pix, g, act);
  } else {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 127) / 128)));
    hipLaunchKernelGGL(conv2d_kernel<4>, grid, dim3(kBlock), 0, s, X, Wt, bias, residual, Y, total_pix, g, act);
  }
}

void pool2d(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int H, int W, int OH, int OW, int kh, int kw,
            int sh, int sw, int pt, int pl, int dh, int dw, bool is_max, bool count_pad) {
  const int64_t total = rows * C * OH * OW;
  if (total <= 0) return;
  hipLaunchKernelGGL(pool2d_kernel, dim3(grid_for(total)), dim3(kBlock), 0, s, X, Y, total, H, W, OH, OW, kh, kw, sh, sw, pt,
                     pl, dh, dw, is_max, count_pad);
}

void global_avgpool(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int S) {
  const int64_t nc = rows * C;
  if (nc <= 0) return;
  hipLaunchKernelGGL(global_avgpool_kernel, dim3(grid_for(nc * 64)), dim3(kBlock), 0, s, X, Y, nc, S);
}

}  // namespace infera_hip::kern
