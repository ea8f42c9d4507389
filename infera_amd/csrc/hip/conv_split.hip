// conv_split.hip -- the tiled convolution on the fp16 matrix cores with every fp32 operand SPLIT in two fp16 halves
// (INFERA_PRECISION=f16x3; BASELINE config C5's MFMA-bound layers).
//
// The exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32) runs at 1/16 of the fp16 / bf16 rate, and ResNet-18 is the one
// configuration whose end-to-end rate is bound by it.  Here every operand v is carried as
//       v * 2^p = hi + lo,     hi = RNE_f16(v * 2^p),   lo = RNE_f16(v * 2^p - hi)          (22 significant bits)
// and every product as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  on three v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3 x 32
// cycles per 16 k-values instead of 8 x 64.  Unlike the bf16 split of mlp_bf16x3.hip (16 significant bits, 2^-17 per product:
// not parity precision) the fp16 split drops only the lo*lo term and the bits below lo -- 2^-22 per product, the size of the
// fp32 rounding the reference's own summation order already differs by -- but fp16 has 5 exponent bits, so the operands must
// be brought into range first.  The power-of-two scales make that exact:
//   * weights: one scale per output feature, chosen at load time so the feature's largest |w| lands in [2^14, 2^15);
//   * activations: one scale PER IMAGE, from the largest |x| of that image's input tensor, which the producing kernel's
//     epilogue tracks (an atomic max per wave; absmax_rows_kernel for tensors produced by other kernels).  Per image, not
//     per batch: a row's result does not depend on which other rows share its pass.
// The epilogue multiplies the accumulator by 2^-(p_image + p_feature) (exact) before bias / residual / activation.
// Elements more than 2^25 below their image's maximum lose relative precision (lo goes subnormal: absolute error 2^-25 of
// the scaled range = 2^-39 of the image's maximum) -- far below what one fp32 rounding of the sum costs.
//
// Geometry, stage order, gathers, LDS weight slabs and the tile mapping are conv2d_tiled_kernel's (conv.hip); only the
// fragment format and the inner product differ.  K-block kb (16 channels) of a 32-channel chunk pairs the gathered quads
// 2kb and 2kb+1: lane (r, h) feeds k = 8h + e  <->  channel 16kb + 8(e >> 2) + 4h + (e & 3) of pixel r.
#include "device_common.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace infera_hip::kern {

namespace {

constexpr int kBlock = 256;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __attribute__((aligned(256))) float g_split_zero_page[128];

__device__ __forceinline__ f16x8 as_h(const u32x4 v) { return __builtin_bit_cast(f16x8, v); }

// two fp32 values (already scaled) -> one dword of fp16 hi halves and one of fp16 lo halves
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned &hi, unsigned &lo) {
  const f32x2 pair = {v0, v1};
  const f16x2 hh = __builtin_convertvector(pair, f16x2);
  const f32x2 rest = {v0 - float(hh[0]), v1 - float(hh[1])};
  hi = __builtin_bit_cast(unsigned, hh);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(rest, f16x2));
}

// scale (a power of two) that brings values of magnitude <= amax (bits of a non-negative float) into [2^14, 2^15),
// and its inverse.  Exponent clamped so both stay normal floats; amax = 0 / tiny -> the largest scale.
__device__ __forceinline__ void scales_of(unsigned amax_bits, float &sc, float &inv) {
  unsigned e = (amax_bits >> 23) & 0xffu;
  e = e < 15u ? 15u : (e > 254u ? 254u : e);
  sc = __uint_as_float((268u - e) << 23);
  inv = __uint_as_float((e - 14u) << 23);
}

// packed split weights: [chunk (conv2d_tiled_pack's stage order)][mt][kb (2)][part (hi, lo)][lane (64)][e (8 halves)]
//   = W[m = 32mt + (lane&31)][tap][c = 32cc + 16kb + 8(e>>2) + 4(lane>>5) + (e&3)] * 2^pw(m)
template <int MT, int S>
__global__ __launch_bounds__(kBlock) void conv2d_split_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                             const float *__restrict__ bias, const float *__restrict__ winv,
                                                             const float *__restrict__ residual, float *__restrict__ Y,
                                                             const unsigned *__restrict__ amax_in, unsigned *__restrict__ amax_out,
                                                             int64_t total_pix, ConvGeom g, ActParam act, unsigned blk0) {
  constexpr int NB = 4 * S;       // gathered quads (16 B per lane) per stage
  constexpr int U = 2 * S * MT;   // units per stage: (chunk, k-block, feature tile) = 2 A fragments + 3 MFMAs
  constexpr int P = 2;            // A-fragment ring depth (units)
  __shared__ __attribute__((aligned(16))) float wbuf[2][S * MT * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const unsigned nfull = gridDim.x & ~7u;
  const unsigned lb = blk0 + (blockIdx.x < nfull ? (blockIdx.x & 7u) * (nfull >> 3) + (blockIdx.x >> 3) : blockIdx.x);
  const int OHW = g.OH * g.OW;
  const int MTtot = g.M / 32, mt0 = blockIdx.y * MT;
  const int CS = g.C / (32 * S), ntaps = g.kh * g.kw, nstages = ntaps * CS;
  const int64_t pix = (int64_t(lb) * 4 + wave) * 32 + r;
  const bool pvalid = pix < total_pix;
  const unsigned pix32 = pvalid ? unsigned(pix) : 0u, n32 = pix32 / unsigned(OHW);
  const int64_t n = n32;
  const int prem = int(pix32 - n32 * unsigned(OHW));
  const int oh = int(unsigned(prem) / unsigned(g.OW)), ow = prem - oh * g.OW;
  const int ih0 = oh * g.sh - g.pt, iw0 = ow * g.sw - g.pl;
  const int HW4 = g.H * g.W * 4;
  const float *xc = X + n * g.H * g.W * g.C + int64_t(h) * HW4 + (int64_t(ih0) * g.W + iw0) * 4;
  const float *zp = g_split_zero_page + 4 * h;
  float sc, sinv;
  scales_of(amax_in[n32], sc, sinv);
  uint64_t okmask = 0;
  if (pvalid) {
    int tap = 0;
    for (int ky = 0; ky < g.kh; ky++)
      for (int kx = 0; kx < g.kw; kx++, tap++) {
        const int iy = ih0 + ky * g.dh, ix = iw0 + kx * g.dw;
        if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) okmask |= uint64_t(1) << tap;
      }
  }

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

  int n_tap = 0, n_kx = 0, n_off = 0, n_base = 0;
  auto gather = [&](f32x4(&b)[NB]) {
    const bool ok = (okmask >> n_tap) & 1;
    const float *p = ok ? xc + n_off : zp;
    const int64_t pstride = ok ? 2 * int64_t(HW4) : 0;
#pragma unroll
    for (int q = 0; q < NB; q++) b[q] = *reinterpret_cast<const f32x4 *>(p + q * pstride);
    n_tap++;
    n_kx++;
    n_off += g.dw * 4;
    if (n_kx == g.kw) {
      n_kx = 0;
      n_off += (g.dh * g.W - g.kw * g.dw) * 4;
    }
    if (n_tap == ntaps) {
      n_tap = 0;
      n_base += 2 * NB * HW4;
      n_off = n_base;
    }
  };
  auto stage_issue = [&](int stage, int buf) {
#pragma unroll
    for (int sl = 0; sl < S; sl++) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(Wp + (int64_t(stage * S + sl) * MTtot + mt0) * 1024) + threadIdx.x;
#pragma unroll
      for (int t = 0; t < MT; t++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + t * 256),
                                         (__attribute__((address_space(3))) void *)(wbuf[buf] + (sl * MT + t) * 1024 + wave * 256), 16, 0, 0);
    }
  };
  auto stage_commit = [] {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  auto step = [&](const f32x4(&bc)[NB], f32x4(&bn)[NB], int stage, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(wbuf[stage & 1]) + lane;
    // unit u -> chunk sl = u / (2 MT), k-block kb = (u / MT) % 2, feature tile t = u % MT; its hi fragment, lo = + 64
    auto fidx = [](int u) { return (((u / (2 * MT)) * MT + u % MT) * 4 + ((u / MT) % 2) * 2) * 64; };
    u32x4 rh[P], rl[P];
#pragma unroll
    for (int u = 0; u < P && u < U; u++) {
      rh[u] = wl[fidx(u)];
      rl[u] = wl[fidx(u) + 64];
    }
    u32x4 bh, bl;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int q = u / MT, t = u % MT;  // q = 2 sl + kb: this unit's gathered quads are 2q and 2q + 1
      if (t == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const f32x4 &src = bc[2 * q + (e >> 1)];
          unsigned hi, lo;
          split_pair(src[2 * (e & 1)] * sc, src[2 * (e & 1) + 1] * sc, hi, lo);
          bh[e] = hi;
          bl[e] = lo;
        }
      }
      const u32x4 ah = rh[u % P], al = rl[u % P];
      if (u + P < U) {
        rh[u % P] = wl[fidx(u + P)];
        rl[u % P] = wl[fidx(u + P) + 64];
      }
      if constexpr (more) {
        if (u == 0) gather(bn);
        if (u == 1) stage_issue(stage + 1, (stage + 1) & 1);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(al), as_h(bh), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah), as_h(bl), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah), as_h(bh), acc[t], 0, 0, 0);
    }
    if constexpr (more) stage_commit();
  };

  f32x4 b0[NB], b1[NB];
  gather(b0);
  stage_issue(0, 0);
  stage_commit();
  constexpr std::true_type kMore{};
  constexpr std::false_type kLast{};
  int stage = 0;
  for (; stage + 2 < nstages; stage += 2) {
    step(b0, b1, stage, kMore);
    step(b1, b0, stage + 1, kMore);
  }
  if (stage + 2 == nstages) {
    step(b0, b1, stage, kMore);
    step(b1, b0, stage + 1, kLast);
  } else {
    step(b0, b1, stage, kLast);
  }

  // epilogue: lane (r,h) holds pixel `pix`, channels 32*(mt0+t) + 8*q + 4h + j -> one 16-byte store per channel quad
  const int64_t OHW4 = int64_t(OHW) * 4;
  const int64_t yoff = n * OHW * g.M + (8 * mt0 + h) * OHW4 + int64_t(prem) * 4;
  float *yp = Y + yoff;
  const float *rp = residual ? residual + yoff : nullptr;
  const f32x4 *bq = bias ? reinterpret_cast<const f32x4 *>(bias + 32 * mt0 + 4 * h) : nullptr;
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(winv + 32 * mt0 + 4 * h);
  float vmax = 0.f;
  auto fetch = [&](f32x4(&bv)[4], f32x4(&rv)[4], f32x4(&wv)[4], int t) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      bv[q] = bq ? bq[8 * t + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
      wv[q] = wq[8 * t + 2 * q];
      rv[q] = (rp && pvalid) ? *reinterpret_cast<const f32x4 *>(rp + (8 * t + 2 * q) * OHW4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  dispatch_act(act.kind, [&](auto kind_tag) {
    constexpr int KIND = decltype(kind_tag)::value;
    f32x4 bv[2][4], rv[2][4], wv[2][4];
    fetch(bv[0], rv[0], wv[0], 0);
#pragma unroll
    for (int t = 0; t < MT; t++) {
      if (t + 1 < MT) fetch(bv[(t + 1) & 1], rv[(t + 1) & 1], wv[(t + 1) & 1], t + 1);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float y = (acc[t][4 * q + j] * wv[t & 1][q][j]) * sinv;
          v[j] = apply_act_c<KIND>((y + bv[t & 1][q][j]) + rv[t & 1][q][j], act.a, act.b);
          vmax = fmaxf(vmax, fabsf(v[j]));
        }
        if (pvalid) *reinterpret_cast<f32x4 *>(yp + (8 * t + 2 * q) * OHW4) = v;
      }
    }
  });
  if (amax_out) {
    // one atomic per wave when its 32 pixels lie in one image (nearly always), one per lane otherwise.  fmaxf drops a NaN
    // operand, so a NaN output is tracked as the other value: the consumer's scale is then arbitrary and its NaN stays NaN.
    const unsigned first = __builtin_amdgcn_readfirstlane(n32);
    const bool uniform = __all(!pvalid || n32 == first);
    if (!pvalid) vmax = 0.f;
    if (uniform) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
      if (lane == 0 && pvalid) atomicMax(amax_out + first, __float_as_uint(vmax));
    } else if (pvalid) {
      atomicMax(amax_out + n32, __float_as_uint(vmax));
    }
  }
}

// per-image max |x| of a tensor some other kernel produced: grid (chunks, rows); bits of a non-negative float, atomic max
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float *__restrict__ X, int64_t per_row, unsigned *__restrict__ amax) {
  const float *x = X + int64_t(blockIdx.y) * per_row;
  const int64_t n4 = per_row >> 2;
  float m = 0.f;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += int64_t(gridDim.x) * 256) {
    const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < per_row; i += 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(amax + blockIdx.y, __float_as_uint(m));
}

// fp32 -> fp16 bits, round to nearest even (host; values are pre-scaled into fp16's range, but every case is handled)
uint16_t f16_bits_rne(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return uint16_t(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));  // NaN / inf
  if (x >= 0x477ff000u) return uint16_t(sign | 0x7c00u);                                   // rounds to >= 65520 -> inf
  if (x < 0x33000001u) return uint16_t(sign);                                              // <= 2^-25 -> 0 (ties to even)
  if (x < 0x38800000u) {  // subnormal half: value = m * 2^-24, m = RNE(f * 2^24)
    const int shift = 126 - int(x >> 23);  // f = 1.mant * 2^(e-127); m = (1.mant * 2^23) >> (shift) with 14 <= shift <= 24
    const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
    const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    return uint16_t(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
  }
  const uint32_t q = ((x - 0x38000000u) >> 13), rem = x & 0x1fffu;  // exponent rebased, 10 mantissa bits kept
  return uint16_t(sign | (q + ((rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ? 1u : 0u)));  // a carry walks into the exponent
}

float f16_bits_to_float(uint16_t hbits) {
  const uint32_t sign = uint32_t(hbits & 0x8000u) << 16, e = (hbits >> 10) & 0x1fu, m = hbits & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else {  // subnormal: m * 2^-24
      float v = float(m) * 5.9604644775390625e-8f;
      std::memcpy(&x, &v, 4);
      x |= sign;
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

}  // namespace

bool conv2d_split_supported(const ConvGeom &g) {
  return conv2d_tiled_supported(g) && g.groups == 1 && g.C % 32 == 0 && g.M % 32 == 0 && g.kvalid == 0 && g.mvalid == 0 && !g.padc;
}

void conv2d_split_pack(const ConvGeom &g, const float *Wt, float *packed, float *winv) {
  const int CC = g.C / 32, MTtot = g.M / 32, ntaps = g.kh * g.kw, S = g.C % 64 == 0 ? 2 : 1;
  const size_t K = size_t(g.C) * ntaps;
  std::vector<float> scale(size_t(g.M));
  for (int m = 0; m < g.M; m++) {
    float amax = 0.f;
    for (size_t k = 0; k < K; k++) {
      const float a = std::fabs(Wt[size_t(m) * K + k]);
      if (a > amax) amax = a;  // (NaN weights: compare false, the feature's outputs are NaN either way)
    }
    uint32_t bits;
    std::memcpy(&bits, &amax, 4);
    uint32_t e = (bits >> 23) & 0xffu;
    e = e < 15u ? 15u : (e > 254u ? 254u : e);
    const uint32_t sb = (268u - e) << 23, ib = (e - 14u) << 23;
    std::memcpy(&scale[size_t(m)], &sb, 4);
    std::memcpy(&winv[m], &ib, 4);
  }
  uint16_t *out = reinterpret_cast<uint16_t *>(packed);
  for (int tap = 0; tap < ntaps; tap++)
    for (int cc = 0; cc < CC; cc++)
      for (int mt = 0; mt < MTtot; mt++)
        for (int kb = 0; kb < 2; kb++)
          for (int lane = 0; lane < 64; lane++)
            for (int e = 0; e < 8; e++) {
              const int m = 32 * mt + (lane & 31), c = 32 * cc + 16 * kb + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
              const size_t chunk = (size_t(cc / S) * ntaps + tap) * S + cc % S;
              const float v = Wt[(size_t(m) * g.C + c) * ntaps + tap] * scale[size_t(m)];
              const uint16_t hi = f16_bits_rne(v);
              const uint16_t lo = f16_bits_rne(v - f16_bits_to_float(hi));
              const size_t base = ((chunk * MTtot + mt) * 2 + kb) * 2;  // units of one fragment = 64 lanes x 8 halves
              out[(base + 0) * 512 + size_t(lane) * 8 + e] = hi;
              out[(base + 1) * 512 + size_t(lane) * 8 + e] = lo;
            }
}

void absmax_rows(hipStream_t s, const float *X, int64_t rows, int64_t per_row, unsigned *amax) {
  if (rows <= 0 || per_row <= 0) return;
  const unsigned chunks = unsigned(std::max<int64_t>(1, std::min<int64_t>(64, per_row / 8192)));
  for (int64_t r0 = 0; r0 < rows; r0 += 65535)
    hipLaunchKernelGGL(absmax_rows_kernel, dim3(chunks, unsigned(std::min<int64_t>(65535, rows - r0))), dim3(256), 0, s, X + r0 * per_row, per_row, amax + r0);
}

void conv2d_split(hipStream_t s, const float *X, const float *packed, const float *bias, const float *winv, const float *residual,
                  float *Y, const unsigned *amax_in, unsigned *amax_out, int64_t rows, const ConvGeom &g, ActParam act) {
  const int64_t total_pix = rows * g.OH * g.OW;
  if (total_pix <= 0) return;
  if (total_pix >= (int64_t(1) << 31)) {
    const int64_t cap = ((int64_t(1) << 31) - 1) / (int64_t(g.OH) * g.OW);
    const int64_t in_row = int64_t(g.C) * g.H * g.W, out_row = int64_t(g.M) * g.OH * g.OW;
    for (int64_t r0 = 0; r0 < rows; r0 += cap)
      conv2d_split(s, X + r0 * in_row, packed, bias, winv, residual ? residual + r0 * out_row : nullptr, Y + r0 * out_row, amax_in + r0,
                   amax_out ? amax_out + r0 : nullptr, std::min(cap, rows - r0), g, act);
    return;
  }
  const unsigned bx = unsigned((total_pix + 127) / 128);
  auto launch = [&](auto kernel, int mt) {
    hipLaunchKernelGGL(kernel, dim3(bx, unsigned(g.M / (32 * mt))), dim3(kBlock), 0, s, X, packed, bias, winv, residual, Y, amax_in, amax_out,
                       total_pix, g, act, 0u);
  };
  const int m32 = g.M / 32;
  const int mt_pick = m32 % 4 == 0 ? 4 : m32 % 3 == 0 ? 3 : m32 % 2 == 0 ? 2 : 1;
  const bool deep = g.C % 64 == 0;
  if (mt_pick == 4) deep ? launch(conv2d_split_kernel<4, 2>, 4) : launch(conv2d_split_kernel<4, 1>, 4);
  else if (mt_pick == 3) deep ? launch(conv2d_split_kernel<3, 2>, 3) : launch(conv2d_split_kernel<3, 1>, 3);
  else if (mt_pick == 2) deep ? launch(conv2d_split_kernel<2, 2>, 2) : launch(conv2d_split_kernel<2, 1>, 2);
  else deep ? launch(conv2d_split_kernel<1, 2>, 1) : launch(conv2d_split_kernel<1, 1>, 1);
}

}  // namespace infera_hip::kern
