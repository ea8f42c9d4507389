// conv_split.hip -- the tiled convolution on the bf16 matrix cores with every fp32 operand cut EXACTLY into three bf16 parts
// (the default form of BASELINE config C5's MFMA-bound layers; INFERA_PRECISION=fp32 selects conv.hip's exact-fp32 kernels instead).
//
// The exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate, and ResNet-18 is the one configuration whose
// end-to-end rate is bound by it.  Here every operand is cut by truncation -- hi = top 16 bits, mid = top 16 bits of
// (v - hi), lo = v - hi - mid: 8 + 8 + 8 = all 24 significant bits, no rounding, no scales, no maxima, and bf16 has fp32's exponent range,
// so nothing about the data has to hold.  A product is six of the nine partial products on v_mfma_f32_32x32x16_bf16 (hi*hi, hi*mid, mid*hi,
// mid*mid, hi*lo, lo*hi; the dropped mid*lo, lo*mid, lo*lo are below 2^-23 of the product), smallest first, fp32 accumulate: 6 x 32 cycles
// per 16 k-values against the exact-fp32 instruction's 8 x 64.
//
// Geometry, stage order, gathers, LDS weight slabs and the tile mapping are conv2d_tiled_kernel's (conv.hip); only the fragment format and
// the inner product differ.  Two kernels: conv2d_split6_kernel<2, TT> for 64-feature slices (compiler-scheduled), conv2d_split6p_kernel for
// 128-feature slices (the same arithmetic with the issue order written down: -3 %, profiles/r04_split6_pipe_ab.txt).  K-block kb (16 channels) of a 32-channel chunk pairs the gathered quads 2kb and 2kb+1: lane (r, h) feeds
// k = 8h + e  <->  channel 16kb + 8(e >> 2) + 4h + (e & 3) of pixel r.
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

__device__ __attribute__((aligned(256))) float g_split_zero_page[128];

using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ bf16x8_t as_b(const u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }

// two fp32 values -> the dwords {part(v0) | part(v1) << 16} of their hi, mid and lo bf16 parts
// An infinite input: hi = inf, mid = inf - inf = NaN -- every output it reaches is NaN, where fp32 arithmetic gives +-inf or NaN depending on
// the weights' signs and zeros.  Passing the infinity through in hi with mid = lo = 0 (two v_cmp_class + two v_cndmask per pair, +36 % of the
// cut's VALU work in the main loop) would not buy fidelity: inf x (a weight's zero mid or lo part) is NaN all the same.  Stated in DESIGN.md 3.3.
// (Round 4 measured and dropped: activations stored PRE-SPLIT by the producing epilogue -- three bf16 planes per 16-channel group, consumers load
// matrix operands and never cut.  Bit-identical results, and SLOWER: 20.97 against 19.59 ms per 1024 ResNet-18 images, every layer class
// (profiles/r04_presplit_ab.txt) -- the cut was never the limiter, the nine-fold tap gathers out of L2 are, and pre-split tensors make them 1.5x
// as many bytes.  The code is in the history: commit f180d80.)
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
// Two stage forms:
//   TT = false: a stage = ONE tap x a 32-channel chunk (two k-blocks); stage order: channel block (one or two chunks) outermost, then the tap, then the
//               chunk inside the block -- conv2d_tiled_pack's chunk order.  Serves every geometry.
//   TT = true (round 4; three-column filters with 64-feature slices, split6_tt()): a stage = the THREE kx taps of one filter row x a 16-channel group (three k-blocks, one per tap); order:
//               channel group outermost, then ky.  The three taps' gathers of a lane are the same cache lines shifted by one pixel (16 bytes) and
//               are issued back to back, so two of the three are served by the L1 instead of coming out of L2 again a stage later -- the tap
//               gathers are what bounds these kernels (profiles/r04_presplit_ab.txt, r04_split6_pixel_tiles_ab.txt).  packed (conv2d_split6_pack):
//               [stage = group * kh + ky][mt][kx (3)][part (hi, mid, lo)][lane (64)][e (8 bf16)].
//
// A SECOND INPUT (round 4; conv2d_split6p_kernel, the kernel every 128-feature launch takes): `x2` -- a tensor of x2.C channels on an H2 x W2 grid, read at (oh * x2.sh, ow * x2.sw) -- supplies
// x2.C / 32 more stages of a 1x1 filter behind the main filter's stages, their weight chunks appended to the blob.  That is a ResNet block's
// projection shortcut computed inside the block's second convolution: out = act(conv3x3(A) + conv1x1/s(P) + (b2 + bd)) in ONE accumulator,
// instead of a separate launch that writes its result and a residual read that fetches it back (x2.X == nullptr: no second input).
template <int MT, bool TT>
__global__ __launch_bounds__(kBlock, MT == 2 ? 3 : 2) void conv2d_split6_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                                 const float *__restrict__ bias, const float *__restrict__ residual,
                                                                 float *__restrict__ Y, int64_t total_pix, ConvGeom g, ActParam act, SecondInput x2) {
  static_assert(MT == 2, "feature tiles per workgroup (four: conv2d_split6p_kernel)");
  constexpr int KB = TT ? 3 : 2;  // k-blocks (16 channels of one tap) per stage
  constexpr int NB = 2 * KB, U = KB * MT, P = 2, SLAB = MT * KB * 768;  // gathered quads per lane and stage; units; A ring depth; floats of weights per stage
  __shared__ __attribute__((aligned(16))) float wbuf[2][SLAB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const unsigned nfull = gridDim.x & ~7u;
  const unsigned lb = blockIdx.x < nfull ? (blockIdx.x & 7u) * (nfull >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int OHW = g.OH * g.OW;
  const int MTtot = g.M / 32, mt0 = blockIdx.y * MT;
  const int CC = g.C / 32, ntaps = g.kh * g.kw, SB = g.C % 64 == 0 ? 2 : 1;
  const int nmain = TT ? (g.C / 16) * g.kh : ntaps * CC, nstages = nmain + (TT || !x2.X ? 0 : x2.C / 32);
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

  // (Round 4 measured one-chunk blocks for the TT = false order -- the nine taps of ONE chunk in consecutive stages: no difference,
  // profiles/r04_split6_stage_order_ab.txt.)
  int n_tap = 0, n_kx = 0, n_off = 0, n_base = 0, n_sl = 0, n_issued = 0;
  // this lane's pixel in the second input (a 1x1 filter, no padding: always inside)
  const int HW4b = x2.H * x2.W * 4;
  const float *xc2 = (!TT && x2.X && pvalid) ? x2.X + n * int64_t(x2.H) * x2.W * x2.C + int64_t(h) * HW4b + (int64_t(oh * x2.sh) * x2.W + ow * x2.sw) * 4 : nullptr;
  auto gather = [&](f32x4(&b)[NB]) {
    if constexpr (!TT) {
      if (n_issued >= nmain) {  // (uniform) a 32-channel chunk of the second input
        const float *p = xc2 ? xc2 + (n_issued - nmain) * (2 * NB * HW4b) : zp;
        const int64_t pstride = xc2 ? 2 * int64_t(HW4b) : 0;
#pragma unroll
        for (int q = 0; q < NB; q++) b[q] = *reinterpret_cast<const f32x4 *>(p + q * pstride);
        n_issued++;
        return;
      }
      n_issued++;
    }
    if constexpr (TT) {
      // the three taps of filter row n_tap (= ky) for channel group n_sl: quads h and h + 2 of the group at each tap
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const bool ok = (okmask >> (n_tap * 3 + kx)) & 1;
        const float *p = ok ? xc + n_off + kx * (g.dw * 4) : zp;
        const int64_t pstride = ok ? 2 * int64_t(HW4) : 0;
        b[2 * kx] = *reinterpret_cast<const f32x4 *>(p);
        b[2 * kx + 1] = *reinterpret_cast<const f32x4 *>(p + pstride);
      }
      n_off += g.dh * g.W * 4;
      if (++n_tap == g.kh) {  // next 16-channel group: four quad planes further on
        n_tap = 0;
        n_base += 4 * HW4;
        n_off = n_base;
      }
      return;
    }
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
    const f32x4 *src = reinterpret_cast<const f32x4 *>(Wp + (int64_t(stage) * MTtot + mt0) * (KB * 768)) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < SLAB / 1024; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * 256),
                                       (__attribute__((address_space(3))) void *)(wbuf[buf] + i * 1024 + wave * 256), 16, 0, 0);
    if constexpr (SLAB % 1024 != 0) {  // (MT = 2, TT: 18 KB = four whole rounds of the workgroup and half a round -- waves 0 and 1)
      if (wave < (SLAB % 1024) / 256)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (SLAB / 1024) * 256),
                                         (__attribute__((address_space(3))) void *)(wbuf[buf] + (SLAB / 1024) * 1024 + wave * 256), 16, 0, 0);
    }
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
    auto fidx = [](int u) { return (((u % MT) * KB + u / MT) * 3) * 64; };  // unit u: k-block u / MT, feature tile u % MT; hi, mid = +64, lo = +128
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
      const u32x4 &bh = bb[kb & 1][0], &bm = bb[kb & 1][1], &bl = bb[kb & 1][2];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(al), as_b(bh), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(ah), as_b(bl), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(am), as_b(bm), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(am), as_b(bh), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(ah), as_b(bm), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(ah), as_b(bh), acc[t], 0, 0, 0);
      // the next k-block's fragments are cut while this one's matrix instructions run
      if (t == 0 && kb + 1 < KB) convert(bc, kb + 1, bb[(kb + 1) & 1][0], bb[(kb + 1) & 1][1], bb[(kb + 1) & 1][2]);
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


// ---- the software-pipelined form of the one-tap stage kernel for 128-feature slices (round 4) ---------------------------------------------------
// Same tiles, stage order, gathers, weight slabs and per-accumulator product order as conv2d_split6_kernel<4, false> (bit-identical results); what
// differs is WHEN things are issued, and that is written down instead of left to the instruction scheduler, which (profiles/r04_split6_pipe_ab.txt)
// read weight fragments from the LDS right in front of the matrix instruction that needs them and put the operand cut between matrix instructions
// on ONE accumulator -- both cost tens of cycles each time (MI355X_MICROARCH.md, measured constants):
//   * a stage is four PAIR-STEPS: (k-block, two feature tiles) = twelve matrix instructions whose accumulators alternate, in three chunks of
//     four closed by sched_barriers; everything else rides in those chunks:
//   * the NEXT pair-step's six weight fragments are read in the first chunk of the current one (two fragment sets of 24 registers);
//   * the operand cut of a k-block is spread over the two pair-steps before its first use;
//   * the barrier of a stage sits in front of its LAST pair-step: by then every wave holds that pair-step's fragments in registers, so the
//     slab can be handed to the DMA of stage s + 2 (two pieces in each of the next three pair-steps), and the next stage's slab -- complete
//     since the wait in front of the barrier -- feeds the fragment reads of stage s + 1's first pair-step without a bubble;
//   * the gather of stage s + 2 is issued in pair-step 2 of stage s (five pair-steps ahead of its first use instead of four), and stays in
//     flight across the barrier: the wait there is vmcnt(<loads of that gather>), not vmcnt(0).
__global__ __launch_bounds__(256, 2) void conv2d_split6p_kernel(const float *__restrict__ X, const float *__restrict__ Wp, const float *__restrict__ bias,
                                                                 const float *__restrict__ residual, float *__restrict__ Y, int64_t total_pix, ConvGeom g,
                                                                 ActParam act, SecondInput x2) {
  constexpr int MT = 4, KB = 2, NB = 4, SLAB = MT * KB * 768, PPW = SLAB / 1024;  // PPW: 1 KiB DMA pieces per wave and slab
  __shared__ __attribute__((aligned(16))) float wbuf[2][SLAB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const unsigned nfull = gridDim.x & ~7u;
  const unsigned lb = blockIdx.x < nfull ? (blockIdx.x & 7u) * (nfull >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int OHW = g.OH * g.OW;
  const int MTtot = g.M / 32, mt0 = blockIdx.y * MT;
  const int CC = g.C / 32, ntaps = g.kh * g.kw, SB = g.C % 64 == 0 ? 2 : 1;
  const int nmain = ntaps * CC, nstages = nmain + (x2.X ? x2.C / 32 : 0);
  const int64_t pix = (int64_t(lb) * 4 + wave) * 32 + r;
  const bool pvalid = pix < total_pix;
  const unsigned pix32 = pvalid ? unsigned(pix) : 0u, n32 = pix32 / unsigned(OHW);
  const int64_t n = n32;
  const int prem = int(pix32 - n32 * unsigned(OHW));
  const int oh = int(unsigned(prem) / unsigned(g.OW)), ow = prem - oh * g.OW;
  const int ih0 = oh * g.sh - g.pt, iw0 = ow * g.sw - g.pl;
  const int HW4 = g.H * g.W * 4;
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

  int n_tap = 0, n_kx = 0, n_off = 0, n_base = 0, n_sl = 0, n_issued = 0;
  const int HW4b = x2.H * x2.W * 4;
  // The next stage's four quads of this lane's pixel (conv2d_split6_kernel's TT = false gather).  Past the main filter's stages: a chunk of the
  // second input; past the last stage (the pipeline issues two stages ahead): the zero page.
  const float *xc = X + n * g.H * g.W * g.C + int64_t(h) * HW4 + (int64_t(ih0) * g.W + iw0) * 4;
  const float *zp = g_split_zero_page + 4 * h;
  const float *xc2 = (x2.X && pvalid) ? x2.X + n * int64_t(x2.H) * x2.W * x2.C + int64_t(h) * HW4b + (int64_t(oh * x2.sh) * x2.W + ow * x2.sw) * 4 : nullptr;
  auto gather = [&](f32x4(&b)[NB]) __attribute__((always_inline)) {
    const bool second = n_issued >= nmain, live2 = xc2 && n_issued < nstages, ok = (okmask >> n_tap) & 1;
    const float *p_main = xc + n_off + n_sl * (2 * NB * HW4), *p_2 = xc2 + (n_issued - nmain) * (2 * NB * HW4b);
    const float *p = second ? (live2 ? p_2 : zp) : (ok ? p_main : zp);
    const int64_t pstride = second ? (live2 ? 2 * int64_t(HW4b) : 0) : (ok ? 2 * int64_t(HW4) : 0);
#pragma unroll
    for (int q = 0; q < NB; q++) b[q] = *reinterpret_cast<const f32x4 *>(p + q * pstride);
    n_issued++;
    n_sl++;
    const bool w1 = n_sl == SB;  // next tap of this channel block
    n_sl = w1 ? 0 : n_sl;
    n_tap += w1;
    n_kx += w1;
    n_off += w1 ? g.dw * 4 : 0;
    const bool w2 = w1 && n_kx == g.kw;  // next filter row
    n_kx = w2 ? 0 : n_kx;
    n_off += w2 ? (g.dh * g.W - g.kw * g.dw) * 4 : 0;
    const bool w3 = w1 && n_tap == ntaps;  // next channel block
    n_tap = w3 ? 0 : n_tap;
    n_base += w3 ? SB * 2 * NB * HW4 : 0;
    n_off = w3 ? n_base : n_off;
  };
  // pieces p0 .. p1 - 1 of this wave's share of the stage's slab -> wbuf[stage & 1] (past the last stage: the last slab again, into a buffer
  // nobody reads any more -- the stage body is the same code for every stage)
  auto dma = [&](int stage, int p0, int p1) __attribute__((always_inline)) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(Wp + (int64_t(min(stage, nstages - 1)) * MTtot + mt0) * (KB * 768)) + threadIdx.x;
    float *dst = wbuf[stage & 1] + wave * 256;
#pragma unroll
    for (int i = p0; i < p1; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * 256),
                                       (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
  };
  // fragments of (k-block kb, feature tiles t0, t0 + 1) of the stage's slab: [tile][hi, mid, lo]
  auto load_a = [&](u32x4(&A)[2][3], int stage, int kb, int t0) __attribute__((always_inline)) {
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(wbuf[stage & 1]) + lane;
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int k = 0; k < 3; k++) A[tt][k] = wl[((t0 + tt) * KB + kb) * 192 + 64 * k];
  };
  // elements e0 .. e1 - 1 (pairs of channels) of k-block kb's three fragments
  // (the empty asm pins the cut to its chunk: the results "are used" there -- without it the optimizer sinks the arithmetic to the first real
  //  use, a pair-step or two later and behind the loads issued in between, which then have to be waited for)
  auto cut = [&](const f32x4(&bc)[NB], int kb, int e0, int e1, u32x4(&o)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = e0; e < e1; e++) {
      const f32x4 &src = bc[2 * kb + (e >> 1)];
      unsigned a, b, c;
      split3_pair(src[2 * (e & 1)], src[2 * (e & 1) + 1], a, b, c);
      asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
      o[0][e] = a;
      o[1][e] = b;
      o[2][e] = c;
    }
  };
#define INFERA_MF(t, a, b) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(a), as_b(b), acc[t], 0, 0, 0)
  // twelve matrix instructions of (fragments A, operand bb) into acc[T0], acc[T0 + 1], in three chunks; f1 / f2 / f3 ride in them
  auto pair_step = [&](auto t0_tag, const u32x4(&A)[2][3], const u32x4(&bb)[3], auto f1, auto f2, auto f3) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0_tag)::value;
    INFERA_MF(T0, A[0][2], bb[0]);
    INFERA_MF(T0 + 1, A[1][2], bb[0]);
    INFERA_MF(T0, A[0][0], bb[2]);
    INFERA_MF(T0 + 1, A[1][0], bb[2]);
    f1();
    __builtin_amdgcn_sched_barrier(0);
    INFERA_MF(T0, A[0][1], bb[1]);
    INFERA_MF(T0 + 1, A[1][1], bb[1]);
    INFERA_MF(T0, A[0][1], bb[0]);
    INFERA_MF(T0 + 1, A[1][1], bb[0]);
    f2();
    __builtin_amdgcn_sched_barrier(0);
    INFERA_MF(T0, A[0][0], bb[1]);
    INFERA_MF(T0 + 1, A[1][0], bb[1]);
    INFERA_MF(T0, A[0][0], bb[0]);
    INFERA_MF(T0 + 1, A[1][0], bb[0]);
    f3();
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr std::integral_constant<int, 0> kT0{};
  constexpr std::integral_constant<int, 2> kT2{};

  f32x4 ba[NB], bq[NB];
  u32x4 bb0[3], bb1[3], A0[2][3], A1[2][3];
  gather(ba);
  dma(0, 0, PPW);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  gather(bq);
  dma(1, 0, 2);
  cut(ba, 0, 0, 4, bb0);
  load_a(A0, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);

  // One stage (the same straight-line code for every stage; what reaches past the last stage is harmless, see gather / dma):
  //   pair-step 0  (k-block 0, tiles 0 1)  + fragments (kb 0, tiles 2 3) | half of the cut of k-block 1      | pieces 2 3 of slab s + 1
  //   pair-step 1  (k-block 0, tiles 2 3)  + fragments (kb 1, tiles 0 1) | the other half                    | pieces 4 5
  //   pair-step 2  (k-block 1, tiles 0 1)  + fragments (kb 1, tiles 2 3) | half of the cut of (s + 1, kb 0)  | gather of stage s + 2
  //   wait: this wave's pieces of slab s + 1 (everything but the four loads of that gather) and its fragment reads; barrier
  //   pair-step 3  (k-block 1, tiles 2 3)  + fragments (s + 1, kb 0, tiles 0 1) | the other half             | pieces 0 1 of slab s + 2
  auto stage_body = [&](f32x4(&bc)[NB], f32x4(&bn)[NB], int s) __attribute__((always_inline)) {
    pair_step(kT0, A0, bb0, [&] { load_a(A1, s, 0, 2); }, [&] { cut(bc, 1, 0, 2, bb1); }, [&] { dma(s + 1, 2, 4); });
    pair_step(kT2, A1, bb0, [&] { load_a(A0, s, 1, 0); }, [&] { cut(bc, 1, 2, 4, bb1); }, [&] { dma(s + 1, 4, 6); });
    pair_step(kT0, A0, bb1, [&] { load_a(A1, s, 1, 2); }, [&] { cut(bn, 0, 0, 2, bb0); }, [&] { gather(bc); });
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pair_step(kT2, A1, bb1, [&] { load_a(A0, s + 1, 0, 0); }, [&] { cut(bn, 0, 2, 4, bb0); }, [&] { dma(s + 2, 0, 2); });
  };
  static_assert(NB == 4 && PPW == 6, "the counter wait in front of the barrier and the piece schedule are written for these");
  int s = 0;
  for (; s + 1 < nstages; s += 2) {
    stage_body(ba, bq, s);
    stage_body(bq, ba, s + 1);
  }
  if (s < nstages) stage_body(ba, bq, s);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the pieces and gathers issued past the last stage)

  if (!pvalid) return;
  const int64_t OHW4 = int64_t(OHW) * 4;
  const int64_t yoff = n * OHW * g.M + (8 * mt0 + h) * OHW4 + int64_t(prem) * 4;
  float *yp = Y + yoff;
  const float *rp = residual ? residual + yoff : nullptr;
  const f32x4 *bqp = bias ? reinterpret_cast<const f32x4 *>(bias + 32 * mt0 + 4 * h) : nullptr;
  dispatch_act(act.kind, [&](auto kind_tag) {
    constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
    for (int t = 0; t < MT; t++) {
      f32x4 bv[4], rv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        bv[q] = bqp ? bqp[8 * t + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
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
#undef INFERA_MF

}  // namespace

// a second input can ride in the one-tap stage form only, which every 128-feature launch takes (conv2d_split6)
bool conv2d_split6_takes_second_input(const ConvGeom &g) { return g.M % 128 == 0; }

bool conv2d_split6_supported(const ConvGeom &g) {
  return conv2d_tiled_supported(g) && g.groups == 1 && g.C % 32 == 0 && g.M % 64 == 0 && g.kvalid == 0 && g.mvalid == 0 && !g.padc;
}

size_t conv2d_split6_packed_floats(const ConvGeom &g) { return size_t(g.kh) * g.kw * g.C * g.M * 3 / 2; }

// Three-column filters whose features come in 64-wide slices take the tap-triple stage form (conv2d_split6_kernel<2, true>): ResNet-18's four
// 64-channel layers 1275 -> 1195 us each (-6 %).  With 128-feature slices a gathered pixel already feeds four feature tiles and the form is 1 %
// SLOWER (36 KB slabs, 232 registers): those stay on the one-tap form (profiles/r04_split6_tap_triple_ab.txt).
static bool split6_tt(const ConvGeom &g) { return g.kw == 3 && g.M % 128 != 0; }

// exact truncation cut of one weight: v = hi + mid + lo, each a bf16
static void cut3(float v, uint16_t &hi, uint16_t &mid, uint16_t &lo) {
  uint32_t x, y, z;
  std::memcpy(&x, &v, 4);
  const uint32_t xh = x & 0xffff0000u;
  float fh, fm;
  std::memcpy(&fh, &xh, 4);
  const float r1 = v - fh;
  std::memcpy(&y, &r1, 4);
  const uint32_t yh = y & 0xffff0000u;
  std::memcpy(&fm, &yh, 4);
  const float r2 = r1 - fm;
  std::memcpy(&z, &r2, 4);
  hi = uint16_t(x >> 16);
  mid = uint16_t(y >> 16);
  lo = uint16_t(z >> 16);
}

void conv2d_split6_pack(const ConvGeom &g, const float *Wt, float *packed) {
  const int CC = g.C / 32, MTtot = g.M / 32, ntaps = g.kh * g.kw, S = g.C % 64 == 0 ? 2 : 1;
  uint16_t *out = reinterpret_cast<uint16_t *>(packed);
  const bool tt = split6_tt(g);
  for (int tap = 0; tap < ntaps; tap++)
    for (int cc = 0; cc < CC; cc++)
      for (int mt = 0; mt < MTtot; mt++)
        for (int kb = 0; kb < 2; kb++)
          for (int lane = 0; lane < 64; lane++)
            for (int e = 0; e < 8; e++) {
              const int m = 32 * mt + (lane & 31), c = 32 * cc + 16 * kb + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
              // fragment group of (tap, 16-channel group 2cc + kb, feature tile mt): three fragments (hi, mid, lo) of 64 lanes x 8 bf16
              size_t base;
              if (tt) {  // [stage = group * kh + ky][mt][kx]
                const int ky = tap / 3, kx = tap % 3;
                base = ((size_t(2 * cc + kb) * g.kh + ky) * MTtot + mt) * 3 + kx;
              } else {  // [chunk (conv2d_tiled_pack's order)][mt][kb]
                const size_t chunk = (size_t(cc / S) * ntaps + tap) * S + cc % S;
                base = (chunk * MTtot + mt) * 2 + kb;
              }
              uint16_t hi, mid, lo;
              cut3(Wt[(size_t(m) * g.C + c) * ntaps + tap], hi, mid, lo);
              out[(base * 3 + 0) * 512 + size_t(lane) * 8 + e] = hi;
              out[(base * 3 + 1) * 512 + size_t(lane) * 8 + e] = mid;
              out[(base * 3 + 2) * 512 + size_t(lane) * 8 + e] = lo;
            }
}

void conv2d_split6(hipStream_t s, const float *X, const float *packed, const float *bias, const float *residual, float *Y, int64_t rows,
                   const ConvGeom &g, ActParam act, const SecondInput &x2) {
  const int64_t total_pix = rows * g.OH * g.OW;
  if (total_pix <= 0) return;
  if (total_pix >= (int64_t(1) << 31)) {
    const int64_t cap = ((int64_t(1) << 31) - 1) / (int64_t(g.OH) * g.OW);
    const int64_t in_row = int64_t(g.C) * g.H * g.W, out_row = int64_t(g.M) * g.OH * g.OW;
    for (int64_t r0 = 0; r0 < rows; r0 += cap)
    {
      SecondInput part = x2;
      if (part.X) part.X += r0 * int64_t(x2.C) * x2.H * x2.W;
      conv2d_split6(s, X + r0 * in_row, packed, bias, residual ? residual + r0 * out_row : nullptr, Y + r0 * out_row, std::min(cap, rows - r0), g, act, part);
    }
    return;
  }
  const unsigned bx = unsigned((total_pix + 127) / 128);
#ifdef INFERA_CONV_PROBES
  // TIMING PROBE ONLY (wrong results): a stride-2 layer's lanes read CONSECUTIVE 16-byte pieces, as they would from an input stored
  // de-interleaved by column phase -- the upper bound of what such a layout could buy (round 6, profiles/r06_stride2_probe.txt)
  ConvGeom gp = g;
  SecondInput x2p = x2;
  if (getenv("INFERA_CONV_PROBE_SW1")) {
    if (gp.sw == 2 && gp.kw == 3) gp.sw = 1;
    if (x2p.X && x2p.sw == 2) x2p.sw = 1;
  }
  auto launch = [&](auto kernel, int features) {
    hipLaunchKernelGGL(kernel, dim3(bx, unsigned(g.M / features)), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, gp, act, x2p);
  };
#else
  auto launch = [&](auto kernel, int features) {
    hipLaunchKernelGGL(kernel, dim3(bx, unsigned(g.M / features)), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, g, act, x2);
  };
#endif
  if (g.M % 128 == 0) launch(conv2d_split6p_kernel, 128);
  else if (split6_tt(g) && !x2.X) launch(conv2d_split6_kernel<2, true>, 64);
  else launch(conv2d_split6_kernel<2, false>, 64);
}

}  // namespace infera_hip::kern
