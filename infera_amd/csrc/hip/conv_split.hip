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
#include "f16_split.hpp"

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

// two fp32 values times the (power-of-two) scale -> one dword of fp16 hi halves and one of fp16 lo halves, five instructions:
// hi = RNE_f16(x * sc) written half by half (v_fma_mixlo / mixhi_f16), the remainders x * sc - hi as exact fp32 FMAs that read
// the fp16 halves in place (v_fma_mix_f32), one packed conversion for lo.  (Plain C++ costs eight: the multiply twice, the hi
// halves widened back by two conversions.)
__device__ __forceinline__ void split_pair(float x0, float x1, float sc, unsigned &hi, unsigned &lo) {
  float r0, r1;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(hi) : "v"(x0), "v"(sc));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hi) : "v"(x1), "v"(sc));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(sc), "v"(hi));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(sc), "v"(hi));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
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
template <int MT, int S, int PROBE = 0>
__global__ __launch_bounds__(kBlock, MT >= 3 ? 2 : 3) void conv2d_split_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
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

  // 16 channels of this lane's pixel (gathered quads 2q and 2q + 1 of the stage) -> the hi and lo B fragments of k-block q
  auto convert = [&](const f32x4(&bc)[NB], int q, u32x4 &oh, u32x4 &ol) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const f32x4 &src = bc[2 * q + (e >> 1)];
      unsigned hi, lo;
      if constexpr (PROBE == 1) {  // (timing probes, PROBES builds only: wrong results)
        hi = __float_as_uint(src[2 * (e & 1)]);
        lo = __float_as_uint(src[2 * (e & 1) + 1]);
      } else {
        split_pair(src[2 * (e & 1)], src[2 * (e & 1) + 1], sc, hi, lo);
      }
      oh[e] = hi;
      ol[e] = lo;
    }
    // The hazard recogniser does not see VALU writes inside inline asm: a matrix instruction issued within two wait states of the last
    // one reads the register's OLD contents (gfx90a+: "VALU write VGPR -> MFMA read", normally padded by the compiler).  Found as a 2^-12
    // per-product error in exactly the instantiations whose first MFMA follows the split directly (S = 1 with one or two feature tiles).
    asm volatile("s_nop 1" : "+v"(oh), "+v"(ol));
  };
  // (Measured and dropped: a PATCH form for stride-1 layers -- per 32-channel chunk the workgroup's whole receptive field parked in LDS already
  // split and scaled (rows of the zero-padded input, a tap = a slot offset, eight fragment planes, every element fetched and split once instead
  // of once per tap), weight slabs per (tap, chunk) stage as here.  Correct (within 1.5e-6 of the oracle, images of different scale sharing a
  // tile) and slower: 128 / 256 / 512-channel layers 0.68-0.77 ms against 0.59-0.72, 64-channel layers 1.20-1.28 against 0.81-0.92.  With the
  // patch in LDS (up to 59 KB) only one 32-channel slab pair fits beside it at two workgroups per CU, so a stage is 12-24 matrix instructions
  // -- half of this form's -- and the per-stage barrier + slab latency cost more than the nine-fold gathers and splits they replaced.  A second
  // version -- eight waves on 256 pixels, TWO taps per stage (this form's 16 MT units between barriers, one slab pair for eight waves) -- ran at
  // 2.2 GHz instead of 2.0 (less VALU and L1 work per matrix instruction) and still only tied: 0.65-0.76 ms on the 128 / 256 / 512-channel layers,
  // 1.05-1.14 on the 64-channel ones.  Counters: 22 % of its LDS cycles are bank conflicts (a wave's 32 pixels wrap over padded rows, so their
  // 16-byte slots are no longer 32 consecutive ones) and the per-tile item tables cost three integer divisions per item -- as much VALU work per
  // tile as the 64-channel layers' whole matrix stream.)
  // (Tuning variants measured neutral to -4 % and removed again: a three-deep A-fragment ring; sched_group_barrier pinning of the split's VALU
  // work between the matrix instructions; s_setprio around them; the weight slab issued before the gathers with `s_waitcnt vmcnt(NB)` at the end
  // of the stage, so that the gathers stay in flight across the barrier.)
  // (Measured and dropped: gathers TWO stages ahead through a third register buffer, the weight slab issued first and `s_waitcnt vmcnt(NB)` at
  // the end of a stage -- 128-feature tiles then need 256 registers and spill 28, 64-feature tiles fall from 3 to 2 waves per SIMD: 4-20 %
  // slower.  With three 32-cycle matrix instructions per product the chip is at its power limit long before the matrix pipe is full -- 1.8 to
  // 2.0 GHz at 41 % busy, against 2.2 GHz at 87 % under the exact-fp32 instruction.)
  auto step = [&](const f32x4(&bc)[NB], f32x4(&bn)[NB], int stage, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    constexpr bool probe_gather = PROBE != 2 && PROBE != 4, probe_stage = PROBE != 3 && PROBE != 4;
    constexpr int Q = 2 * S;  // k-blocks per stage
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(wbuf[stage & 1]) + lane;
    // unit u -> k-block q = u / MT (chunk q / 2, half q % 2), feature tile t = u % MT; its hi fragment, lo = + 64
    auto fidx = [](int u) { return ((((u / MT) / 2) * MT + u % MT) * 4 + ((u / MT) % 2) * 2) * 64; };
    u32x4 rh[P], rl[P];
#pragma unroll
    for (int u = 0; u < P && u < U; u++) {
      rh[u] = wl[fidx(u)];
      rl[u] = wl[fidx(u) + 64];
    }
    u32x4 bh[2], bl[2];
    convert(bc, 0, bh[0], bl[0]);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int q = u / MT, t = u % MT;
      const u32x4 ah = rh[u % P], al = rl[u % P];
      if (u + P < U) {
        rh[u % P] = wl[fidx(u + P)];
        rl[u % P] = wl[fidx(u + P) + 64];
      }
      if constexpr (more) {
        if constexpr (probe_gather) {
          if (u == 0) gather(bn);
        }
        if constexpr (probe_stage) {
          if (u == 1) stage_issue(stage + 1, (stage + 1) & 1);
        }
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(al), as_h(bh[q & 1]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah), as_h(bl[q & 1]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah), as_h(bh[q & 1]), acc[t], 0, 0, 0);
      if constexpr (PROBE == 6) {  // (timing probe: six matrix instructions per unit -- what a three-part bf16 split would issue)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(al), as_h(bl[q & 1]), acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(al), as_h(bh[q & 1]), acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah), as_h(bl[q & 1]), acc[t], 0, 0, 0);
      }
      // the next k-block's fragments are split while this one's matrix instructions run
      if (t == 0 && q + 1 < Q) convert(bc, q + 1, bh[(q + 1) & 1], bl[(q + 1) & 1]);
    }
    if constexpr (more && probe_stage) stage_commit();
    if constexpr (!probe_gather) {
#pragma unroll
      for (int q = 0; q < NB; q++) bn[q] = bc[q];
    }
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
  dispatch_act(act.kind, [&](auto kind_tag) {
    constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
    for (int t = 0; t < MT; t++) {
      f32x4 bv[4], rv[4], wv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        bv[q] = bq ? bq[8 * t + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
        wv[q] = wq[8 * t + 2 * q];
        rv[q] = (rp && pvalid) ? *reinterpret_cast<const f32x4 *>(rp + (8 * t + 2 * q) * OHW4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float y = (acc[t][4 * q + j] * wv[q][j]) * sinv;
          v[j] = apply_act_c<KIND>((y + bv[q][j]) + rv[q][j], act.a, act.b);
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

// ---- weight-stationary persistent form (conv2d_ws_kernel's structure, conv.hip) ------------------------------------------------
// When one M-slice of the packed split weights fits in LDS (the 64-channel 3x3 layers, the 64 -> 128 stride-2 entry in two 64-feature
// slices, every 1x1 downsample) a persistent workgroup loads it once and each of its NW waves walks 32-pixel tiles on its own: no weight
// slab per stage, no barrier after the prologue, the stage stream running through tile boundaries (the first stage of the wave's next
// tile is gathered under the last stage of this one, the epilogue's loads / stores / the max-tracking atomic drain under the next tile).
// The activation scale belongs to the tile being COMPUTED: the prefetch cursor loads its image's maximum when it enters a tile, and the
// tile takes it over when it starts (the cursor is then exactly one stage into that tile).
template <int MT, int S, int NW>
__global__ __launch_bounds__(NW * 64) void conv2d_split_ws_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                                 const float *__restrict__ bias, const float *__restrict__ winv,
                                                                 const float *__restrict__ residual, float *__restrict__ Y,
                                                                 const unsigned *__restrict__ amax_in, unsigned *__restrict__ amax_out,
                                                                 int64_t total_pix, ConvGeom g, ActParam act) {
  constexpr int NB = 4 * S, U = 2 * S * MT, P = 2, Q = 2 * S;
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int OHW = g.OH * g.OW;
  const int MTtot = g.M / 32, mt0 = blockIdx.y * MT;
  const int CS = g.C / (32 * S), ntaps = g.kh * g.kw, nstages = ntaps * CS, nchunks = nstages * S;
  for (int i = threadIdx.x; i < nchunks * MT * 256; i += NW * 64) {
    const int chunk = i / (MT * 256), rem = i - chunk * (MT * 256);
    reinterpret_cast<f32x4 *>(wlds)[i] = reinterpret_cast<const f32x4 *>(Wp + (int64_t(chunk) * MTtot + mt0) * 1024)[rem];
  }
  __syncthreads();

  const int64_t ntiles = (total_pix + 31) >> 5;
  const int64_t tstride = int64_t(gridDim.x) * NW;
  int64_t tile = int64_t(blockIdx.x) * NW + wave;
  if (tile >= ntiles) return;
  const int HW4 = g.H * g.W * 4;
  const float *zp = g_split_zero_page + 4 * h;

  const float *p_xc = zp;
  uint64_t p_ok = 0;
  // image maxima of the last two tiles the cursor has entered: the tile that starts computing takes a_new, or a_old when a tile is a
  // single stage (the cursor is then a whole tile further on: it entered the next tile while gathering this one's only stage)
  unsigned a_new = 0, a_old = 0;
  int p_tap = 0, p_kx = 0, p_off = 0, p_base = 0;
  auto enter_tile = [&](int64_t t) {
    const int64_t pix = (t << 5) + r;
    const bool pvalid = t < ntiles && pix < total_pix;
    const unsigned pix32 = pvalid ? unsigned(pix) : 0u, n32 = pix32 / unsigned(OHW);
    const int64_t n = n32;
    const int prem = int(pix32 - n32 * unsigned(OHW));
    const int oh = int(unsigned(prem) / unsigned(g.OW)), ow = prem - oh * g.OW;
    const int ih0 = oh * g.sh - g.pt, iw0 = ow * g.sw - g.pl;
    p_xc = X + n * int64_t(g.H) * g.W * g.C + int64_t(h) * HW4 + (int64_t(ih0) * g.W + iw0) * 4;
    a_old = a_new;
    a_new = amax_in[n32];
    p_ok = 0;
    if (pvalid) {
      int tap = 0;
      for (int ky = 0; ky < g.kh; ky++)
        for (int kx = 0; kx < g.kw; kx++, tap++) {
          const int iy = ih0 + ky * g.dh, ix = iw0 + kx * g.dw;
          if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) p_ok |= uint64_t(1) << tap;
        }
    }
    p_tap = p_kx = p_off = p_base = 0;
  };
  int64_t p_tile = tile;
  auto gather = [&](f32x4(&b)[NB]) {
    const bool ok = (p_ok >> p_tap) & 1;
    const float *p = ok ? p_xc + p_off : zp;
    const int64_t pstride = ok ? 2 * int64_t(HW4) : 0;
#pragma unroll
    for (int q = 0; q < NB; q++) b[q] = *reinterpret_cast<const f32x4 *>(p + q * pstride);
    p_tap++;
    p_kx++;
    p_off += g.dw * 4;
    if (p_kx == g.kw) {
      p_kx = 0;
      p_off += (g.dh * g.W - g.kw * g.dw) * 4;
    }
    if (p_tap == ntaps) {
      p_tap = 0;
      p_base += 2 * NB * HW4;
      p_off = p_base;
      if (p_base == CS * 2 * NB * HW4) {
        p_tile += tstride;
        enter_tile(p_tile);
      }
    }
  };

  f32x16 acc[MT];
  float sc = 1.f, sinv = 1.f;
  auto convert = [&](const f32x4(&bc)[NB], int q, u32x4 &oh, u32x4 &ol) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const f32x4 &src = bc[2 * q + (e >> 1)];
      unsigned hi, lo;
      split_pair(src[2 * (e & 1)], src[2 * (e & 1) + 1], sc, hi, lo);
      oh[e] = hi;
      ol[e] = lo;
    }
    // The hazard recogniser does not see VALU writes inside inline asm: a matrix instruction issued within two wait states of the last
    // one reads the register's OLD contents (gfx90a+: "VALU write VGPR -> MFMA read", normally padded by the compiler).  Found as a 2^-12
    // per-product error in exactly the instantiations whose first MFMA follows the split directly (S = 1 with one or two feature tiles).
    asm volatile("s_nop 1" : "+v"(oh), "+v"(ol));
  };
  auto step = [&](const f32x4(&bc)[NB], f32x4(&bn)[NB], int stage) {
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(wlds + int64_t(stage) * S * MT * 1024) + lane;
    auto fidx = [](int u) { return ((((u / MT) / 2) * MT + u % MT) * 4 + ((u / MT) % 2) * 2) * 64; };
    u32x4 rh[P], rl[P];
#pragma unroll
    for (int u = 0; u < P && u < U; u++) {
      rh[u] = wl[fidx(u)];
      rl[u] = wl[fidx(u) + 64];
    }
    u32x4 bh[2], bl[2];
    convert(bc, 0, bh[0], bl[0]);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int q = u / MT, t = u % MT;
      const u32x4 ah = rh[u % P], al = rl[u % P];
      if (u + P < U) {
        rh[u % P] = wl[fidx(u + P)];
        rl[u % P] = wl[fidx(u + P) + 64];
      }
      if (u == 0) gather(bn);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(al), as_h(bh[q & 1]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah), as_h(bl[q & 1]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah), as_h(bh[q & 1]), acc[t], 0, 0, 0);
      if (t == 0 && q + 1 < Q) convert(bc, q + 1, bh[(q + 1) & 1], bl[(q + 1) & 1]);
    }
  };
  const f32x4 *bq = bias ? reinterpret_cast<const f32x4 *>(bias + 32 * mt0 + 4 * h) : nullptr;
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(winv + 32 * mt0 + 4 * h);
  const int64_t OHW4 = int64_t(OHW) * 4;
  auto epilogue = [&](int64_t t) {
    const int64_t pix = (t << 5) + r;
    const bool pvalid = pix < total_pix;
    const unsigned n32 = (pvalid ? unsigned(pix) : 0u) / unsigned(OHW);
    const int64_t n = n32;
    const int prem = int((pvalid ? unsigned(pix) : 0u) - n32 * unsigned(OHW));
    const int64_t yoff = n * OHW * int64_t(g.M) + int64_t(8 * mt0 + h) * OHW4 + int64_t(prem) * 4;
    float *yp = Y + yoff;
    const float *rp = residual ? residual + yoff : nullptr;
    float vmax = 0.f;
    dispatch_act(act.kind, [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
      for (int t2 = 0; t2 < MT; t2++) {
        f32x4 bv[4], rv[4], wv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          bv[q] = bq ? bq[8 * t2 + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
          wv[q] = wq[8 * t2 + 2 * q];
          rv[q] = (rp && pvalid) ? *reinterpret_cast<const f32x4 *>(rp + (8 * t2 + 2 * q) * OHW4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float y = (acc[t2][4 * q + j] * wv[q][j]) * sinv;
            v[j] = apply_act_c<KIND>((y + bv[q][j]) + rv[q][j], act.a, act.b);
            vmax = fmaxf(vmax, fabsf(v[j]));
          }
          if (pvalid) *reinterpret_cast<f32x4 *>(yp + (8 * t2 + 2 * q) * OHW4) = v;
        }
      }
    });
    if (amax_out) {
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
  };
  auto run_tile = [&](f32x4(&ba)[NB], f32x4(&bb)[NB], int64_t t) {
    scales_of(nstages == 1 ? a_old : a_new, sc, sinv);
#pragma unroll
    for (int t2 = 0; t2 < MT; t2++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t2][i] = 0.f;
    int stage = 0;
    for (; stage + 2 <= nstages; stage += 2) {
      step(ba, bb, stage);
      step(bb, ba, stage + 1);
    }
    if (stage < nstages) step(ba, bb, stage);
    epilogue(t);
  };

  f32x4 b0[NB], b1[NB];
  enter_tile(tile);
  gather(b0);
  const bool odd = nstages & 1;
  for (;;) {
    run_tile(b0, b1, tile);
    tile += tstride;
    if (tile >= ntiles) break;
    if (odd) {
      run_tile(b1, b0, tile);
      tile += tstride;
      if (tile >= ntiles) break;
    }
  }
}

// ---- bf16 x 3 parts: the precondition-free sibling (INFERA_PRECISION=bf16x6) -------------------------------------------------------
// Every fp32 operand is cut EXACTLY into three bf16 parts by truncation -- hi = top 16 bits, mid = top 16 bits of (v - hi), lo = v - hi - mid:
// 8 + 8 + 8 = all 24 significant bits, no rounding, no scales, no maxima, and bf16 has fp32's exponent range, so nothing about the data has to
// hold.  A product is six of the nine partial products on v_mfma_f32_32x32x16_bf16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the dropped
// mid*lo, lo*mid, lo*lo are below 2^-23 of the product), smallest first, fp32 accumulate: 6 x 32 cycles per 16 k-values against the exact-fp32
// instruction's 8 x 64.  Tiled geometry with ONE 32-channel chunk per stage (a chunk's fragments are 6 KB per 32 features: 2 k-blocks x 3 parts),
// 64 or 128 features per workgroup; everything else -- gathers, tile order, epilogue without any scaling -- is conv2d_tiled_kernel's.
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ bf16x8_t as_b(const u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }

// two fp32 values -> the dwords {part(v0) | part(v1) << 16} of their hi, mid and lo bf16 parts
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned &hi, unsigned &mid, unsigned &lo) {
  const unsigned x0 = __float_as_uint(v0), x1 = __float_as_uint(v1);
  const float r0 = v0 - __uint_as_float(x0 & 0xffff0000u), r1 = v1 - __uint_as_float(x1 & 0xffff0000u);  // exact: 16 bits left
  const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(y0 & 0xffff0000u), s1 = r1 - __uint_as_float(y1 & 0xffff0000u);  // exact: 8 bits left
  hi = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
  mid = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// packed: [chunk (conv2d_tiled_pack's order)][mt][kb (2)][part (hi, mid, lo)][lane (64)][e (8 bf16)]
template <int MT>
__global__ __launch_bounds__(kBlock, 2) void conv2d_split6_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                                 const float *__restrict__ bias, const float *__restrict__ residual,
                                                                 float *__restrict__ Y, int64_t total_pix, ConvGeom g, ActParam act) {
  static_assert(MT == 2 || MT == 4, "a stage's fragments (MT x 6 KB) are whole 4 KB pieces of the workgroup's copy");
  constexpr int NB = 4, U = 2 * MT, P = 2, SLAB = MT * 1536;  // floats per stage
  __shared__ __attribute__((aligned(16))) float wbuf[2][SLAB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const unsigned nfull = gridDim.x & ~7u;
  const unsigned lb = blockIdx.x < nfull ? (blockIdx.x & 7u) * (nfull >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int OHW = g.OH * g.OW;
  const int MTtot = g.M / 32, mt0 = blockIdx.y * MT;
  const int CC = g.C / 32, ntaps = g.kh * g.kw, nstages = ntaps * CC, SB = g.C % 64 == 0 ? 2 : 1;
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

  // stage order: channel block (SB chunks) outermost, then the tap, then the chunk inside the block -- conv2d_tiled_pack's chunk order, one
  // chunk per stage
  int n_tap = 0, n_kx = 0, n_off = 0, n_base = 0, n_sl = 0;
  auto gather = [&](f32x4(&b)[NB]) {
    const bool ok = (okmask >> n_tap) & 1;
    const float *p = ok ? xc + n_off + n_sl * (2 * NB * HW4) : zp;
    const int64_t pstride = ok ? 2 * int64_t(HW4) : 0;
#pragma unroll
    for (int q = 0; q < NB; q++) b[q] = *reinterpret_cast<const f32x4 *>(p + q * pstride);
    if (++n_sl == SB) {  // next tap of this channel block
      n_sl = 0;
      n_tap++;
      n_kx++;
      n_off += g.dw * 4;
      if (n_kx == g.kw) {
        n_kx = 0;
        n_off += (g.dh * g.W - g.kw * g.dw) * 4;
      }
      if (n_tap == ntaps) {
        n_tap = 0;
        n_base += SB * 2 * NB * HW4;
        n_off = n_base;
      }
    }
  };
  auto stage_issue = [&](int stage, int buf) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(Wp + (int64_t(stage) * MTtot + mt0) * 1536) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < SLAB / 1024; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * 256),
                                       (__attribute__((address_space(3))) void *)(wbuf[buf] + i * 1024 + wave * 256), 16, 0, 0);
  };
  auto stage_commit = [] {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  auto convert = [&](const f32x4(&bc)[NB], int kb, u32x4 &oh, u32x4 &om, u32x4 &ol) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const f32x4 &src = bc[2 * kb + (e >> 1)];
      unsigned a, b, c;
      split3_pair(src[2 * (e & 1)], src[2 * (e & 1) + 1], a, b, c);
      oh[e] = a;
      om[e] = b;
      ol[e] = c;
    }
  };
  auto step = [&](const f32x4(&bc)[NB], f32x4(&bn)[NB], int stage, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(wbuf[stage & 1]) + lane;
    auto fidx = [](int u) { return (((u % MT) * 2 + u / MT) * 3) * 64; };  // unit u: k-block u / MT, feature tile u % MT; hi, mid = +64, lo = +128
    u32x4 ra[P][3], bb[2][3];
#pragma unroll
    for (int u = 0; u < P && u < U; u++)
#pragma unroll
      for (int k = 0; k < 3; k++) ra[u][k] = wl[fidx(u) + 64 * k];
    convert(bc, 0, bb[0][0], bb[0][1], bb[0][2]);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int kb = u / MT, t = u % MT;
      const u32x4 ah = ra[u % P][0], am = ra[u % P][1], al = ra[u % P][2];
      if (u + P < U) {
#pragma unroll
        for (int k = 0; k < 3; k++) ra[u % P][k] = wl[fidx(u + P) + 64 * k];
      }
      if constexpr (more) {
        if (u == 0) gather(bn);
        if (u == 1) stage_issue(stage + 1, (stage + 1) & 1);
      }
      const u32x4 &bh = bb[kb][0], &bm = bb[kb][1], &bl = bb[kb][2];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(al), as_b(bh), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(ah), as_b(bl), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(am), as_b(bm), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(am), as_b(bh), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(ah), as_b(bm), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(ah), as_b(bh), acc[t], 0, 0, 0);
      if (t == 0 && kb == 0) convert(bc, 1, bb[1][0], bb[1][1], bb[1][2]);
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

  if (!pvalid) return;
  const int64_t OHW4 = int64_t(OHW) * 4;
  const int64_t yoff = n * OHW * g.M + (8 * mt0 + h) * OHW4 + int64_t(prem) * 4;
  float *yp = Y + yoff;
  const float *rp = residual ? residual + yoff : nullptr;
  const f32x4 *bq = bias ? reinterpret_cast<const f32x4 *>(bias + 32 * mt0 + 4 * h) : nullptr;
  dispatch_act(act.kind, [&](auto kind_tag) {
    constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
    for (int t = 0; t < MT; t++) {
      f32x4 bv[4], rv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        bv[q] = bq ? bq[8 * t + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
        rv[q] = rp ? *reinterpret_cast<const f32x4 *>(rp + (8 * t + 2 * q) * OHW4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = apply_act_c<KIND>((acc[t][4 * q + j] + bv[q][j]) + rv[q][j], act.a, act.b);
        *reinterpret_cast<f32x4 *>(yp + (8 * t + 2 * q) * OHW4) = v;
      }
    }
  });
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
    uint32_t sb, ib;
    f16_split_scale_bits(amax, sb, ib);
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

// ---- bf16 x 3 parts (conv2d_split6_kernel) ----
bool conv2d_split6_supported(const ConvGeom &g) { return conv2d_split_supported(g) && g.M % 64 == 0; }

size_t conv2d_split6_packed_floats(const ConvGeom &g) { return size_t(g.kh) * g.kw * g.C * g.M * 3 / 2; }

void conv2d_split6_pack(const ConvGeom &g, const float *Wt, float *packed) {
  const int CC = g.C / 32, MTtot = g.M / 32, ntaps = g.kh * g.kw, S = g.C % 64 == 0 ? 2 : 1;
  uint16_t *out = reinterpret_cast<uint16_t *>(packed);
  for (int tap = 0; tap < ntaps; tap++)
    for (int cc = 0; cc < CC; cc++)
      for (int mt = 0; mt < MTtot; mt++)
        for (int kb = 0; kb < 2; kb++)
          for (int lane = 0; lane < 64; lane++)
            for (int e = 0; e < 8; e++) {
              const int m = 32 * mt + (lane & 31), c = 32 * cc + 16 * kb + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
              const size_t chunk = (size_t(cc / S) * ntaps + tap) * S + cc % S;
              const float v = Wt[(size_t(m) * g.C + c) * ntaps + tap];
              uint32_t x, y, z;  // exact truncation split: v = hi + mid + lo
              std::memcpy(&x, &v, 4);
              const uint32_t xh = x & 0xffff0000u;
              float fh;
              std::memcpy(&fh, &xh, 4);
              const float r1 = v - fh;
              std::memcpy(&y, &r1, 4);
              const uint32_t yh = y & 0xffff0000u;
              float fm;
              std::memcpy(&fm, &yh, 4);
              const float r2 = r1 - fm;
              std::memcpy(&z, &r2, 4);
              const size_t base = ((chunk * MTtot + mt) * 2 + kb) * 3;  // fragments of 64 lanes x 8 bf16
              out[(base + 0) * 512 + size_t(lane) * 8 + e] = uint16_t(x >> 16);
              out[(base + 1) * 512 + size_t(lane) * 8 + e] = uint16_t(y >> 16);
              out[(base + 2) * 512 + size_t(lane) * 8 + e] = uint16_t(z >> 16);
            }
}

void conv2d_split6(hipStream_t s, const float *X, const float *packed, const float *bias, const float *residual, float *Y, int64_t rows,
                   const ConvGeom &g, ActParam act) {
  const int64_t total_pix = rows * g.OH * g.OW;
  if (total_pix <= 0) return;
  if (total_pix >= (int64_t(1) << 31)) {
    const int64_t cap = ((int64_t(1) << 31) - 1) / (int64_t(g.OH) * g.OW);
    const int64_t in_row = int64_t(g.C) * g.H * g.W, out_row = int64_t(g.M) * g.OH * g.OW;
    for (int64_t r0 = 0; r0 < rows; r0 += cap)
      conv2d_split6(s, X + r0 * in_row, packed, bias, residual ? residual + r0 * out_row : nullptr, Y + r0 * out_row, std::min(cap, rows - r0), g, act);
    return;
  }
  const unsigned bx = unsigned((total_pix + 127) / 128);
  if (g.M % 128 == 0)
    hipLaunchKernelGGL((conv2d_split6_kernel<4>), dim3(bx, unsigned(g.M / 128)), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, g, act);
  else
    hipLaunchKernelGGL((conv2d_split6_kernel<2>), dim3(bx, unsigned(g.M / 64)), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, g, act);
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
#ifdef INFERA_CONV_PROBES
  // 1 no operand split, 2 no gathers, 3 no weight staging / barrier, 4 neither (bare matrix stream), 6 six matrix instructions per unit (a three-part bf16 split's count): timing only, results are wrong
  static const int probe = getenv("INFERA_SPLIT_PROBE") ? atoi(getenv("INFERA_SPLIT_PROBE")) : 0;
  if (probe && deep && (mt_pick == 4 || mt_pick == 2)) {
    switch (probe * 2 + (mt_pick == 4)) {
      case 2: return launch(conv2d_split_kernel<2, 2, 1>, 2);
      case 3: return launch(conv2d_split_kernel<4, 2, 1>, 4);
      case 4: return launch(conv2d_split_kernel<2, 2, 2>, 2);
      case 5: return launch(conv2d_split_kernel<4, 2, 2>, 4);
      case 6: return launch(conv2d_split_kernel<2, 2, 3>, 2);
      case 7: return launch(conv2d_split_kernel<4, 2, 3>, 4);
      case 8: return launch(conv2d_split_kernel<2, 2, 4>, 2);
      case 9: return launch(conv2d_split_kernel<4, 2, 4>, 4);
      case 12: return launch(conv2d_split_kernel<2, 2, 6>, 2);
      case 13: return launch(conv2d_split_kernel<4, 2, 6>, 4);
    }
  }
#endif
  // weight-stationary persistent form when one M-slice of the split weights fits in LDS and the launch fills the persistent grid
  // (INFERA_CONV_WS as for the exact-fp32 kernels: 0 never, 1 default, 2 whenever it fits -- read per launch, tests compare both)
  const char *ws_env = getenv("INFERA_CONV_WS");
  const int ws_mode = ws_env ? atoi(ws_env) : 1;
  if (ws_mode == 2 || (ws_mode == 1 && total_pix >= 32 * 2048)) {
    constexpr size_t kWsLdsBytes = 160 * 1024 - 256;
    const size_t slice32 = size_t(g.kh) * g.kw * g.C * 32 * sizeof(float);  // hi + lo fp16 fragments of 32 features: as many bytes as fp32
    auto launch_ws = [&](auto kernel, int mt, int nw) {
      static std::atomic<uint64_t> attr_done[4] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      static int cus[64] = {};
      if (!cus[dev & 63]) (void)hipDeviceGetAttribute(&cus[dev & 63], hipDeviceAttributeMultiprocessorCount, dev);
      std::atomic<uint64_t> &done = attr_done[(mt == 4 ? 2 : 0) + (deep ? 1 : 0)];
      if (!((done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(kWsLdsBytes));
        done.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
      }
      const unsigned slices = unsigned(g.M / (32 * mt));
      const int64_t ntiles = (total_pix + 31) / 32;
      unsigned gx = unsigned(std::max(1, cus[dev & 63] / int(slices)));
      gx = unsigned(std::min<int64_t>(gx, (ntiles + nw - 1) / nw));
      hipLaunchKernelGGL(kernel, dim3(gx, slices), dim3(unsigned(nw) * 64), slice32 * mt, s, X, packed, bias, winv, residual, Y, amax_in, amax_out,
                         total_pix, g, act);
    };
    // (128-feature slices -- the 1x1 downsamples -- only when forced: 256 registers, 22 spilled, 260 us against the tiled form's 234)
    if (ws_mode == 2 && m32 % 4 == 0 && slice32 * 4 <= kWsLdsBytes)
      return deep ? launch_ws(conv2d_split_ws_kernel<4, 2, 8>, 4, 8) : launch_ws(conv2d_split_ws_kernel<4, 1, 8>, 4, 8);
    // (1x1 layers stay on the tiled form unless forced: a single stage per tile, 298 us against 234 for the 64 -> 128 downsample)
    if (m32 % 2 == 0 && slice32 * 2 <= kWsLdsBytes && (g.kh * g.kw > 1 || ws_mode == 2)) return deep ? launch_ws(conv2d_split_ws_kernel<2, 2, 8>, 2, 8) : launch_ws(conv2d_split_ws_kernel<2, 1, 8>, 2, 8);
  }
  // (Measured and dropped: conv2d_tiled's tail split -- the last partial round of a 128-feature launch as a second launch of 32- or 64-feature
  // tiles.  Every workgroup here gathers and splits its whole input whatever its feature count, so a quarter of the features costs nearly a
  // whole workgroup time: 567 + 212 us (32-feature tail) or 580 + 72 ... 646 + 155 us (64-feature tail) against 640-705 unsplit.)
  if (mt_pick == 4) deep ? launch(conv2d_split_kernel<4, 2>, 4) : launch(conv2d_split_kernel<4, 1>, 4);
  else if (mt_pick == 3) deep ? launch(conv2d_split_kernel<3, 2>, 3) : launch(conv2d_split_kernel<3, 1>, 3);
  else if (mt_pick == 2) deep ? launch(conv2d_split_kernel<2, 2>, 2) : launch(conv2d_split_kernel<2, 1>, 2);
  else deep ? launch(conv2d_split_kernel<1, 2>, 1) : launch(conv2d_split_kernel<1, 1>, 1);
}

}  // namespace infera_hip::kern
