// conv.hip -- 2-D convolution / pooling kernels for the BLOB (image) path (BASELINE config C5).
//
// Convolution is an implicit GEMM on the exact-fp32 matrix cores, in the same transposed formulation
// as dense.hip / mlp_fused.hip:
//       Out^T[m, p] = sum_k  Wt[m, k] * col[k, p],     p = (n, oh, ow) output pixel,  k = filter tap x channel
//   A operand = weights      (lane: m = lane&31, k-pair by lane half)
//   B operand = im2col gather of the input, computed on the fly per lane -- never materialised in HBM.
//
// Two kernels:
//   * conv2d_tiled_kernel  -- the workhorse (every ResNet layer except the stem).  Activations are kept
//     CHANNELS-LAST (NHWC) inside a conv plan, and K is ordered (ky, kx, c), so the 4 k-values a lane
//     feeds to 4 consecutive MFMA k-steps are 4 consecutive channels of ONE input pixel: one 16-byte load
//     per lane per 8 k (bounds-checked once per filter tap), and one 16-byte store per lane per 4 output
//     channels.  A workgroup = 4 waves = 128 output pixels x (MT*32) output channels; weight fragments
//     (pre-packed fragment-major at load time) are staged through LDS in 32-channel chunks,
//     double-buffered, one barrier per chunk, and shared by the 4 waves; the next chunk's B operands are
//     in flight while the current chunk's 16*MT MFMAs run.  Bias (BatchNormalization already folded in by
//     the loader) + activation are fused in the epilogue.
//   * conv2d_generic_kernel -- any geometry / groups / layouts (the C=3 stem reads the caller's NCHW
//     blob and writes NHWC); scalar gathers, weights straight from L2.
#include "device_common.hpp"

namespace infera_hip::kern {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int64_t act_index(bool nhwc, int64_t n, int c, int y, int x, int C, int H, int W) {
  return nhwc ? ((n * H + y) * W + x) * C + c : ((n * C + c) * H + y) * int64_t(W) + x;
}

// Generic kernel.  Wk = weights transposed at load time to [group][k][Mg] (k = (c, ky, kx) as in ONNX), so
// the A-operand load is coalesced over the 32 output channels of a tile.  The per-k input offset and
// (ky, kx) come from a table built in LDS once per workgroup -- no integer division in the k loop.
struct KEntry {
  int off;       // input offset of tap k relative to the (ih0, iw0) corner of the receptive field
  short ky, kx;  // dilated tap coordinates, for the bounds test
};

template <int MT>
__global__ __launch_bounds__(kBlock) void conv2d_generic_kernel(const float *__restrict__ X, const float *__restrict__ Wk,
                                                               const float *__restrict__ bias, float *__restrict__ Y,
                                                               int64_t total_pix, ConvGeom g, ActParam act, bool in_nhwc,
                                                               bool out_nhwc) {
  extern __shared__ __attribute__((aligned(16))) KEntry ktab[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int Cg = g.C / g.groups, Mg = g.M / g.groups;
  const int khw = g.kh * g.kw, KK = Cg * khw;
  const int OHW = g.OH * g.OW;
  const int mtiles = (Mg + 32 * MT - 1) / (32 * MT);
  const int grp = blockIdx.y / mtiles, mt0 = (blockIdx.y % mtiles) * 32 * MT;
  for (int k = threadIdx.x; k < KK; k += kBlock) {
    const int c = k / khw, rem = k - c * khw, ky = rem / g.kw, kx = rem - ky * g.kw;
    const int cy = ky * g.dh, cx = kx * g.dw, cc = grp * Cg + c;
    ktab[k].off = in_nhwc ? (cy * g.W + cx) * g.C + cc : (cc * g.H + cy) * g.W + cx;
    ktab[k].ky = short(cy);
    ktab[k].kx = short(cx);
  }
  __syncthreads();
  const int64_t pix = (int64_t(blockIdx.x) * 4 + wave) * 32 + r;
  const bool pvalid = pix < total_pix;
  const int64_t n = pvalid ? pix / OHW : 0;
  const int prem = pvalid ? int(pix % OHW) : 0;
  const int oh = prem / g.OW, ow = prem % g.OW;
  const int ih0 = oh * g.sh - g.pt, iw0 = ow * g.sw - g.pl;
  // corner of the receptive field (may lie outside the image; only in-bounds taps are dereferenced)
  const float *xc = X + n * int64_t(g.C) * g.H * g.W + (in_nhwc ? (int64_t(ih0) * g.W + iw0) * g.C : int64_t(ih0) * g.W + iw0);
  const float *wg = Wk + int64_t(grp) * KK * Mg + mt0 + r;

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

  for (int k0 = 0; k0 < KK; k0 += 2) {
    const int k = k0 + h;
    float b = 0.f;
    float a[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) a[t] = 0.f;
    if (k < KK) {
      const KEntry e = ktab[k];
      const int iy = ih0 + e.ky, ix = iw0 + e.kx;
      if (pvalid && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) b = xc[e.off];
#pragma unroll
      for (int t = 0; t < MT; t++)
        if (mt0 + 32 * t + r < Mg) a[t] = wg[int64_t(k) * Mg + 32 * t];
    }
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b, acc[t], 0, 0, 0);
  }
  if (!pvalid) return;
  // lane (r,h) holds pixel `pix`, channels mt0 + 32t + 8*q + 4h + j
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ml = mt0 + 32 * t + 8 * q + 4 * h;
      if (out_nhwc && ml + 3 < Mg && (g.M & 3) == 0) {  // one 16-byte NHWC store per channel quad
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = apply_act(acc[t][4 * q + j] + (bias ? bias[grp * Mg + ml + j] : 0.f), act);
        *reinterpret_cast<f32x4 *>(Y + pix * g.M + grp * Mg + ml) = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (ml + j < Mg) {
            const int m = grp * Mg + ml + j;
            const float v = acc[t][4 * q + j] + (bias ? bias[m] : 0.f);
            Y[act_index(out_nhwc, n, m, oh, ow, g.M, g.OH, g.OW)] = apply_act(v, act);
          }
      }
    }
}

// ---- tiled NHWC kernel --------------------------------------------------------------------------------------
// packed weights: [chunk = tap*(C/32) + cc][mt (all M/32 tiles)][g (4)][lane (64)][j (4)]
//   = Wt[m = 32mt + (lane&31)][tap][c = 32cc + 8g + 4*(lane>>5) + j]
template <int MT>
__global__ __launch_bounds__(kBlock) void conv2d_tiled_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                             const float *__restrict__ bias, const float *__restrict__ residual,
                                                             float *__restrict__ Y, int64_t total_pix, ConvGeom g, ActParam act) {
  __shared__ __attribute__((aligned(16))) float wbuf[2][MT * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int OHW = g.OH * g.OW;
  const int MTtot = g.M / 32, mt0 = blockIdx.y * MT;
  const int CC = g.C / 32, ntaps = g.kh * g.kw, nchunks = ntaps * CC;
  const int64_t pix = (int64_t(blockIdx.x) * 4 + wave) * 32 + r;
  const bool pvalid = pix < total_pix;
  const int64_t n = pvalid ? pix / OHW : 0;
  const int prem = pvalid ? int(pix % OHW) : 0;
  const int oh = prem / g.OW, ow = prem % g.OW;
  const int ih0 = oh * g.sh - g.pt, iw0 = ow * g.sw - g.pl;
  const float *xn = X + n * int64_t(g.H) * g.W * g.C + 4 * h;

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

  // B operands of one chunk: 4 groups x 16 bytes per lane (zeros outside the image / past the table)
  auto gather = [&](f32x4(&b)[4], int chunk) {
    const int tap = chunk / CC, cc = chunk - tap * CC;
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    const int iy = ih0 + ky * g.dh, ix = iw0 + kx * g.dw;
    const bool ok = pvalid && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(xn + (int64_t(iy) * g.W + ix) * g.C + cc * 32);
#pragma unroll
    for (int q = 0; q < 4; q++) b[q] = ok ? p[2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // this block's MT tiles of one chunk are contiguous in the packed blob: MT*1024 floats, MT float4 per thread
  auto stage_load = [&](f32x4(&wreg)[MT], int chunk) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(Wp + (int64_t(chunk) * MTtot + mt0) * 1024) + threadIdx.x;
#pragma unroll
    for (int t = 0; t < MT; t++) wreg[t] = src[t * 256];
  };
  auto stage_store = [&](const f32x4(&wreg)[MT], int buf) {
    f32x4 *dst = reinterpret_cast<f32x4 *>(wbuf[buf]) + threadIdx.x;
#pragma unroll
    for (int t = 0; t < MT; t++) dst[t * 256] = wreg[t];
  };

  f32x4 bcur[4], bnext[4], wreg[MT];
  gather(bcur, 0);
  stage_load(wreg, 0);
  stage_store(wreg, 0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; chunk++) {
    const bool more = chunk + 1 < nchunks;
    if (more) {
      gather(bnext, chunk + 1);
      stage_load(wreg, chunk + 1);
    }
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(wbuf[chunk & 1]) + lane;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      f32x4 a[MT];
#pragma unroll
      for (int t = 0; t < MT; t++) a[t] = wl[(t * 4 + q) * 64];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], bcur[q][j], acc[t], 0, 0, 0);
    }
    if (more) {
      stage_store(wreg, (chunk + 1) & 1);
#pragma unroll
      for (int q = 0; q < 4; q++) bcur[q] = bnext[q];
    }
    __syncthreads();
  }
  if (!pvalid) return;
  // epilogue: lane (r,h) holds pixel `pix`, channels 32*(mt0+t) + 8*q + 4h + j -> one 16-byte NHWC store per quad
  // (optional residual: the block's skip tensor, same NHWC layout -- the Add of a ResNet block is fused here)
  float *yp = Y + pix * g.M + 32 * mt0 + 4 * h;
  const float *rp = residual ? residual + pix * g.M + 32 * mt0 + 4 * h : nullptr;
  const float *bp = bias ? bias + 32 * mt0 + 4 * h : nullptr;
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      f32x4 v, res = {0.f, 0.f, 0.f, 0.f};
      if (rp) res = *reinterpret_cast<const f32x4 *>(rp + 32 * t + 8 * q);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float x = acc[t][4 * q + j] + (bp ? bp[32 * t + 8 * q + j] : 0.f);
        if (rp) x += res[j];
        v[j] = apply_act(x, act);
      }
      *reinterpret_cast<f32x4 *>(yp + 32 * t + 8 * q) = v;
    }
}

// NHWC pooling, 4 channels (16 bytes) per thread: consecutive lanes walk the channel axis, so every tap
// is a fully coalesced read and the output a coalesced 16-byte store.
__global__ __launch_bounds__(kBlock) void pool2d_nhwc4_kernel(const float *__restrict__ X, float *__restrict__ Y, int64_t total4,
                                                             int C4, int H, int W, int OH, int OW, int kh, int kw, int sh,
                                                             int sw, int pt, int pl, int dh, int dw, bool is_max, bool count_pad) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  const f32x4 *x4 = reinterpret_cast<const f32x4 *>(X);
  f32x4 *y4 = reinterpret_cast<f32x4 *>(Y);
  for (int64_t o = int64_t(blockIdx.x) * kBlock + threadIdx.x; o < total4; o += stride) {
    const int c4 = int(o % C4);
    const int ow = int((o / C4) % OW);
    const int oh = int((o / (int64_t(C4) * OW)) % OH);
    const int64_t n = o / (int64_t(C4) * OW * OH);
    f32x4 acc = is_max ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < kh; i++) {
      const int iy = oh * sh - pt + i * dh;
      if (iy < 0 || iy >= H) continue;
      for (int j = 0; j < kw; j++) {
        const int ix = ow * sw - pl + j * dw;
        if (ix < 0 || ix >= W) continue;
        const f32x4 v = x4[((n * H + iy) * W + ix) * C4 + c4];
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] = is_max ? fmaxf(acc[e], v[e]) : acc[e] + v[e];
        cnt++;
      }
    }
    if (!is_max) {
      const float d = float(count_pad ? kh * kw : (cnt ? cnt : 1));
#pragma unroll
      for (int e = 0; e < 4; e++) acc[e] = acc[e] / d;
    }
    y4[o] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void pool2d_kernel(const float *__restrict__ X, float *__restrict__ Y, int64_t total,
                                                       int C, int H, int W, int OH, int OW, int kh, int kw, int sh, int sw,
                                                       int pt, int pl, int dh, int dw, bool is_max, bool count_pad, bool nhwc) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t o = int64_t(blockIdx.x) * kBlock + threadIdx.x; o < total; o += stride) {
    int c, oh, ow;
    int64_t n;
    if (nhwc) {
      c = int(o % C);
      ow = int((o / C) % OW);
      oh = int((o / (int64_t(C) * OW)) % OH);
      n = o / (int64_t(C) * OW * OH);
    } else {
      ow = int(o % OW);
      oh = int((o / OW) % OH);
      c = int((o / (int64_t(OW) * OH)) % C);
      n = o / (int64_t(OW) * OH * C);
    }
    float acc = is_max ? -INFINITY : 0.f;
    int cnt = 0;
    for (int i = 0; i < kh; i++)
      for (int j = 0; j < kw; j++) {
        const int iy = oh * sh - pt + i * dh, ix = ow * sw - pl + j * dw;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float v = X[act_index(nhwc, n, c, iy, ix, C, H, W)];
        acc = is_max ? fmaxf(acc, v) : acc + v;
        cnt++;
      }
    if (!is_max) acc = acc / float(count_pad ? kh * kw : (cnt ? cnt : 1));
    Y[o] = acc;
  }
}

// NCHW: one wave per (n, c) over S contiguous elements.  NHWC: one lane per (n, c), stride C.
__global__ __launch_bounds__(kBlock) void global_avgpool_kernel(const float *__restrict__ X, float *__restrict__ Y,
                                                               int64_t nc_total, int C, int S, bool nhwc) {
  if (nhwc) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t o = int64_t(blockIdx.x) * kBlock + threadIdx.x; o < nc_total; o += stride) {
      const int64_t n = o / C;
      const int c = int(o % C);
      const float *src = X + n * int64_t(S) * C + c;
      float acc = 0.f;
      for (int i = 0; i < S; i++) acc += src[int64_t(i) * C];
      Y[o] = acc / float(S);
    }
    return;
  }
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

bool conv2d_generic_supported(const ConvGeom &g) {
  return size_t(g.C / g.groups) * g.kh * g.kw * sizeof(KEntry) <= 64 * 1024;  // the k table lives in LDS
}

void conv2d_generic_pack(const ConvGeom &g, const float *Wt, float *packed) {
  const int Mg = g.M / g.groups, KK = (g.C / g.groups) * g.kh * g.kw;
  for (int grp = 0; grp < g.groups; grp++)
    for (int m = 0; m < Mg; m++)
      for (int k = 0; k < KK; k++) packed[(size_t(grp) * KK + k) * Mg + m] = Wt[(size_t(grp) * Mg + m) * KK + k];
}

void conv2d(hipStream_t s, const float *X, const float *Wk, const float *bias, float *Y, int64_t rows, const ConvGeom &g,
            ActParam act, bool in_nhwc, bool out_nhwc) {
  const int64_t total_pix = rows * g.OH * g.OW;
  if (total_pix <= 0) return;
  const int Mg = g.M / g.groups;
  const size_t lds = size_t(g.C / g.groups) * g.kh * g.kw * sizeof(KEntry);
  const unsigned bx = unsigned((total_pix + 127) / 128);
  if (Mg <= 32) {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 31) / 32)));
    hipLaunchKernelGGL(conv2d_generic_kernel<1>, grid, dim3(kBlock), lds, s, X, Wk, bias, Y, total_pix, g, act, in_nhwc, out_nhwc);
  } else if (Mg <= 64) {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 63) / 64)));
    hipLaunchKernelGGL(conv2d_generic_kernel<2>, grid, dim3(kBlock), lds, s, X, Wk, bias, Y, total_pix, g, act, in_nhwc, out_nhwc);
  } else {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 127) / 128)));
    hipLaunchKernelGGL(conv2d_generic_kernel<4>, grid, dim3(kBlock), lds, s, X, Wk, bias, Y, total_pix, g, act, in_nhwc, out_nhwc);
  }
}

bool conv2d_tiled_supported(const ConvGeom &g) { return g.groups == 1 && g.C % 32 == 0 && g.M % 64 == 0; }

size_t conv2d_tiled_packed_floats(const ConvGeom &g) { return size_t(g.kh) * g.kw * g.C * g.M; }

void conv2d_tiled_pack(const ConvGeom &g, const float *Wt, float *packed) {
  const int CC = g.C / 32, MTtot = g.M / 32, ntaps = g.kh * g.kw;
  for (int tap = 0; tap < ntaps; tap++)
    for (int cc = 0; cc < CC; cc++)
      for (int mt = 0; mt < MTtot; mt++)
        for (int q = 0; q < 4; q++)
          for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 4; j++) {
              const int m = 32 * mt + (lane & 31), c = 32 * cc + 8 * q + 4 * (lane >> 5) + j;
              const size_t chunk = size_t(tap) * CC + cc;
              packed[((chunk * MTtot + mt) * 4 + q) * 256 + size_t(lane) * 4 + j] = Wt[(size_t(m) * g.C + c) * ntaps + tap];
            }
}

void conv2d_tiled(hipStream_t s, const float *X, const float *packed, const float *bias, const float *residual, float *Y,
                  int64_t rows, const ConvGeom &g, ActParam act) {
  const int64_t total_pix = rows * g.OH * g.OW;
  if (total_pix <= 0) return;
  const unsigned bx = unsigned((total_pix + 127) / 128);
  if (g.M % 128 == 0) {
    hipLaunchKernelGGL(conv2d_tiled_kernel<4>, dim3(bx, unsigned(g.M / 128)), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, g, act);
  } else {
    hipLaunchKernelGGL(conv2d_tiled_kernel<2>, dim3(bx, unsigned(g.M / 64)), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, g, act);
  }
}

void pool2d(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int H, int W, int OH, int OW, int kh, int kw,
            int sh, int sw, int pt, int pl, int dh, int dw, bool is_max, bool count_pad, bool nhwc) {
  const int64_t total = rows * C * OH * OW;
  if (total <= 0) return;
  if (nhwc && C % 4 == 0) {
    hipLaunchKernelGGL(pool2d_nhwc4_kernel, dim3(grid_for(total / 4)), dim3(kBlock), 0, s, X, Y, total / 4, C / 4, H, W, OH, OW, kh, kw,
                       sh, sw, pt, pl, dh, dw, is_max, count_pad);
    return;
  }
  hipLaunchKernelGGL(pool2d_kernel, dim3(grid_for(total)), dim3(kBlock), 0, s, X, Y, total, C, H, W, OH, OW, kh, kw, sh, sw, pt,
                     pl, dh, dw, is_max, count_pad, nhwc);
}

void global_avgpool(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int S, bool nhwc) {
  const int64_t nc = rows * C;
  if (nc <= 0) return;
  hipLaunchKernelGGL(global_avgpool_kernel, dim3(grid_for(nhwc ? nc : nc * 64)), dim3(kBlock), 0, s, X, Y, nc, C, S, nhwc);
}

}  // namespace infera_hip::kern
