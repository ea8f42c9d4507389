// conv.hip -- 2-D convolution / pooling kernels for the BLOB (image) path (BASELINE config C5).
//
// Convolution is an implicit GEMM on the exact-fp32 matrix cores, in the same transposed formulation
// as dense.hip / mlp_fused.hip:
//       Out^T[m, p] = sum_k  Wt[m, k] * col[k, p],     p = (n, oh, ow) output pixel,  k = filter tap x channel
//   A operand = weights      (lane: m = lane&31, k-pair by lane half)
//   B operand = im2col gather of the input, computed on the fly per lane -- never materialised in HBM.
//
// Two kernels:
// Inside a conv plan activations live in CHANNEL-QUAD PLANES ("CQ"): [N][C/4][H][W][4] -- the 4 k-values a
// lane feeds to 4 consecutive MFMA k-steps are one 16-byte load, the 4 output channels a lane holds per
// accumulator quad are one 16-byte store, and -- unlike NHWC -- the 32 pixels of a wave (lane&31) are 32
// CONSECUTIVE 16-byte pieces of a plane row, so every gather and every store is a coalesced 512-byte run
// (NHWC measured 64 L1 accesses per gather instruction, CQ 16).
//   * conv2d_tiled_kernel  -- the workhorse (every ResNet layer except the stem), K ordered (ky, kx, c).
//     A workgroup = 4 waves = 128 output pixels x (MT*32) output channels; weight fragments (pre-packed
//     fragment-major at load time) are staged through LDS, double-buffered, one barrier per stage, shared
//     by the 4 waves; the next stage's B operands are in flight while the current stage's MFMAs run.
//     Bias (BatchNormalization already folded in by the loader), the residual Add of a ResNet block and
//     the activation are fused in the epilogue.
//   * conv2d_generic_kernel -- any geometry / groups / layouts (the C=3 stem reads the caller's NCHW
//     blob and writes CQ); scalar gathers, weights straight from L2.
#include "device_common.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>

namespace infera_hip::kern {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int64_t act_index(bool cq, int64_t n, int c, int y, int x, int C, int H, int W) {
  return cq ? (((n * (C >> 2) + (c >> 2)) * H + y) * W + x) * 4 + (c & 3) : ((n * C + c) * H + y) * int64_t(W) + x;
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
                                                               int64_t total_pix, ConvGeom g, ActParam act, bool in_cq,
                                                               bool out_cq) {
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
    ktab[k].off = in_cq ? (((cc >> 2) * g.H + cy) * g.W + cx) * 4 + (cc & 3) : (cc * g.H + cy) * g.W + cx;
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
  const float *xc = X + n * int64_t(g.C) * g.H * g.W + (int64_t(ih0) * g.W + iw0) * (in_cq ? 4 : 1);
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
  // bias first (all loads in flight together), activation resolved once, then the stores
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int ml = mt0 + 32 * t + 8 * (i >> 2) + 4 * h + (i & 3);
      acc[t][i] += (bias && ml < Mg) ? bias[grp * Mg + ml] : 0.f;
    }
  dispatch_act(act.kind, [&](auto kind_tag) {
    constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t][i] = apply_act_c<KIND>(acc[t][i], act.a, act.b);
  });
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ml = mt0 + 32 * t + 8 * q + 4 * h;
      if (out_cq && ml + 3 < Mg && (Mg & 3) == 0) {  // one 16-byte store per channel quad, coalesced over the wave's pixels
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = acc[t][4 * q + j];
        *reinterpret_cast<f32x4 *>(Y + act_index(true, n, grp * Mg + ml, oh, ow, g.M, g.OH, g.OW)) = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (ml + j < Mg) Y[act_index(out_cq, n, grp * Mg + ml + j, oh, ow, g.M, g.OH, g.OW)] = acc[t][4 * q + j];
      }
    }
}

// ---- tiled NHWC kernel --------------------------------------------------------------------------------------
// packed weights: [chunk = ((cc/S)*taps + tap)*S + cc%S][mt (all M/32 tiles)][g (4)][lane (64)][j (4)]
//   = Wt[m = 32mt + (lane&31)][tap][c = 32cc + 8g + 4*(lane>>5) + j]
//
// One LDS stage = S consecutive 32-channel chunks of one filter tap (S = 2 when C % 64 == 0): S*MT KB of
// fragments shared by the 4 waves (loaded global -> LDS directly), double-buffered, ONE barrier per stage (= per
// 16*S*MT MFMAs per wave).
// Inside a stage the wave runs the same "unit" pipeline as mlp_device.inc: a unit = one 16-byte A fragment
// (ds_read_b128, P units ahead in a register ring) + 4 MFMAs, pinned with sched_barrier so the loads stay
// interleaved with the matrix stream.  The next stage's B operands (S*4 16-byte gathers per lane) and weight
// slab are requested in the first units and land under the remaining MFMAs.  Out-of-image taps do not
// branch: a per-lane bit mask (one bit per tap, built once) redirects the gather to a page of zeros.
// (Measured and dropped: a persistent variant walking a tile list with the pipeline running through the
// tile boundary -- same time at 25k tiles, worse tails at <= 1.5k tiles; 64-feature tiles for the deep
// layers; static s_setprio staggering of co-resident workgroups; gathers two stages ahead through a third
// B buffer for the 64-feature tiles -- 2 instead of 3 waves per SIMD, same time; 8-wave workgroups sharing one
// weight slab (half the weight stream from L2) -- 7 % slower.  A bare kernel (no loads, no stores, same MFMA
// stream and index math) runs the 128-channel layer at 152 TFLOP/s: the matrix cores are not the limit, the
// memory operations around them are.  Also neutral: non-temporal output stores; s_setprio 3 during the prologue
// and epilogue.  Probes: tools/conv_probe.sh.)
__device__ __attribute__((aligned(256))) float g_zero_page[128];
#ifdef INFERA_CONV_PROBES
// [MT==4][phase]: summed shader cycles per wave: prologue, main loop, epilogue issue, store drain; [4] = waves
__device__ unsigned long long g_conv_stamps[2][12];
#endif

// DENSE: a fully connected layer as a 1x1 convolution over 1x1 images (the channel-quad layout of a 1x1 image is the
// row-major matrix).  g.C / g.M are padded to multiples of 32; the input rows are g.kvalid floats long (a multiple
// of 4: whole quads past the end read the zero page), the output rows g.mvalid (any length: stores are guarded).
// MODE 2 (PADC): a convolution whose channel counts are whole quads but no multiples of 32: g.C / g.M are padded as for
// DENSE (zero weights / bias), the tensors hold g.kvalid / g.mvalid channels -- channel-quad planes past the real ones
// read the zero page on the way in and are not stored on the way out.
template <int MT, int S, int PROBE = 0, int MODE = 0>
__global__ __launch_bounds__(kBlock) void conv2d_tiled_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                             const float *__restrict__ bias, const float *__restrict__ residual,
                                                             float *__restrict__ Y, int64_t total_pix, ConvGeom g, ActParam act,
                                                             unsigned blk0 = 0) {
  constexpr bool DENSE = MODE == 1, PADC = MODE == 2;
  constexpr int NB = 4 * S;   // B fragments (16 B per lane) per stage
  constexpr int U = NB * MT;  // units per stage
  constexpr int P = 3;        // A-fragment ring depth (2 and 4 measured identical)
  __shared__ __attribute__((aligned(16))) float wbuf[2][S * MT * 1024];
#ifdef INFERA_CONV_PROBES
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, ta = 0, tb = 0, tc = 0, w_vm = 0, w_bar = 0, w_ring = 0;
  if constexpr (PROBE == 5) t0 = __builtin_readcyclecounter();
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a contiguous
  // range of pixel tiles (neighbouring tiles share halo rows -> they share that XCD's L2).
  const unsigned nfull = gridDim.x & ~7u;
  // (blk0: this launch covers pixel blocks [blk0, blk0 + gridDim.x) of the layer -- the tail of a launch split in two, see conv2d_tiled)
  const unsigned lb = blk0 + (blockIdx.x < nfull ? (blockIdx.x & 7u) * (nfull >> 3) + (blockIdx.x >> 3) : blockIdx.x);
  const int OHW = g.OH * g.OW;
  const int MTtot = g.M / 32, mt0 = blockIdx.y * MT;
  const int CS = g.C / (32 * S), ntaps = g.kh * g.kw, nstages = ntaps * CS;
  const int64_t pix = (int64_t(lb) * 4 + wave) * 32 + r;
  const bool pvalid = pix < total_pix;
  // (32-bit: the launcher keeps total_pix below 2^31; a 64-bit division is ~150 instructions per lane)
  const unsigned pix32 = pvalid ? unsigned(pix) : 0u, n32 = pix32 / unsigned(OHW);
  const int64_t n = n32;
  const int prem = int(pix32 - n32 * unsigned(OHW));
  const int oh = int(unsigned(prem) / unsigned(g.OW)), ow = prem - oh * g.OW;
  const int ih0 = oh * g.sh - g.pt, iw0 = ow * g.sw - g.pl;
  // receptive-field corner of this lane's pixel (may lie outside the image; masked taps never dereference it)
  const int HW4 = g.H * g.W * 4;  // floats per channel-quad plane
  const float *xc = DENSE ? X + n * g.kvalid + 4 * h
                          : X + n * g.H * g.W * (PADC ? g.kvalid : g.C) + int64_t(h) * HW4 + (int64_t(ih0) * g.W + iw0) * 4;
  const float *zp = g_zero_page + 4 * h;
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

  // wave-uniform position of the stage being PREFETCHED.  Stage order is CHANNEL-BLOCK major, taps fastest: two
  // consecutive stages read the same channel planes shifted by one tap, i.e. mostly the same cache lines while
  // they are still in L2 (with taps outermost the shifted re-read came C/(32 S) stages -- several MB of other
  // planes per XCD -- later and went back to HBM: 49 GB of reads per 1024-image pass against ~15 GB of tensors).
  int n_tap = 0, n_kx = 0, n_off = 0, n_base = 0, n_plane = 0;
  auto gather = [&](f32x4(&b)[NB]) {
    const bool ok = (okmask >> n_tap) & 1;
    if constexpr (PADC) {
      // group q of this channel block is plane n_plane + 2q + h; planes >= kvalid / 4 do not exist
      const int planes = g.kvalid >> 2;
#pragma unroll
      for (int q = 0; q < NB; q++) {
        const bool in = ok && n_plane + 2 * q + h < planes;
        b[q] = *reinterpret_cast<const f32x4 *>(in ? xc + n_off + q * 2 * int64_t(HW4) : zp);
      }
    } else if constexpr (DENSE) {
      // this lane's quad of group q starts at column n_off + 8q + 4h; columns >= kvalid do not exist
#pragma unroll
      for (int q = 0; q < NB; q++) {
        const bool in = ok && n_off + 8 * q + 4 * h < g.kvalid;
        b[q] = *reinterpret_cast<const f32x4 *>(in ? xc + n_off + 8 * q : zp);
      }
    } else {
      const float *p = ok ? xc + n_off : zp;
      const int64_t pstride = ok ? 2 * int64_t(HW4) : 0;  // group q+1 = two channel-quad planes further
#pragma unroll
      for (int q = 0; q < NB; q++) b[q] = *reinterpret_cast<const f32x4 *>(p + q * pstride);
    }
    // advance to the following stage: next tap of this channel block, else first tap of the next block
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
      n_plane += 2 * NB;
    }
  };
  // This block's MT tiles of one 32-channel chunk are contiguous in the packed blob (MT*1024 floats); they go
  // STRAIGHT into LDS (global_load_lds_dwordx4: each wave drops 1 KB at its wave-uniform base + lane*16), no
  // staging registers and no ds_write pass.  Completion is tracked by vmcnt like any other load.
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
  // all of this wave's outstanding loads (gathers AND LDS-bound weight pieces) have landed, then the workgroup meets
  auto stage_commit = [] {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  // `more` is a compile-time constant per call site: with a run-time flag the prefetch sits in a branch and
  // hipcc's waitcnt pass, merging the two paths, makes every other stage wait on the loads it has just issued.
  auto step = [&](const f32x4(&bc)[NB], f32x4(&bn)[NB], int stage, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(wbuf[stage & 1]) + lane;
    // unit u -> B fragment q = u / MT (slab q/4, group q%4), feature tile t = u % MT
    auto fidx = [](int u) { return (((u / MT) / 4 * MT + u % MT) * 4 + (u / MT) % 4) * 64; };
    f32x4 ring[P];
#pragma unroll
    for (int u = 0; u < P && u < U; u++) ring[u] = wl[fidx(u)];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int q = u / MT, t = u % MT;
      f32x4 a = ring[u % P];
      if constexpr (PROBE == 4) {
        asm volatile("" : "+v"(a));
      } else {
        if (u + P < U) ring[u % P] = wl[fidx(u + P)];
      }
      if constexpr (more) {
        if constexpr (PROBE != 2 && PROBE != 3 && PROBE != 4) {
          if (u == 0) gather(bn);
        }
        if constexpr (PROBE != 3 && PROBE != 4 && PROBE != 6) {
          if (u == 1) stage_issue(stage + 1, (stage + 1) & 1);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bc[q][j], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef INFERA_CONV_PROBES
    if constexpr (more && PROBE == 5) {  // where a stage boundary spends its time
      const unsigned long long s0 = __builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long s1 = __builtin_readcyclecounter();
      const unsigned long long s2 = s1;
      __syncthreads();
      const unsigned long long s3 = __builtin_readcyclecounter();
      w_vm += s1 - s0;
      w_ring += s2 - s1;
      w_bar += s3 - s2;
    }
#endif
    if constexpr (more && (PROBE == 0 || PROBE == 6 || PROBE == 2)) {
      stage_commit();
    }
    if constexpr (PROBE == 2 || PROBE == 3 || PROBE == 4) {
#pragma unroll
      for (int q = 0; q < NB; q++) bn[q] = bc[q];
    }
  };

  f32x4 b0[NB], b1[NB];
#ifdef INFERA_CONV_PROBES
  if constexpr (PROBE == 5) ta = __builtin_readcyclecounter();
#endif
  gather(b0);
  stage_issue(0, 0);
#ifdef INFERA_CONV_PROBES
  if constexpr (PROBE == 5) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tb = __builtin_readcyclecounter();
  }
#endif
  stage_commit();
#ifdef INFERA_CONV_PROBES
  if constexpr (PROBE == 5) t1 = __builtin_readcyclecounter();
#endif
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
#ifdef INFERA_CONV_PROBES
  if constexpr (PROBE == 5) t2 = __builtin_readcyclecounter();
#endif
  if (!pvalid) return;
  // epilogue: lane (r,h) holds pixel `pix`, channels 32*(mt0+t) + 8*q + 4h + j -> one 16-byte NHWC store per quad
  // (optional residual: the block's skip tensor, same NHWC layout -- the Add of a ResNet block is fused here)
  const int64_t OHW4 = int64_t(OHW) * 4;
  const int64_t yoff = DENSE ? n * g.mvalid + (8 * mt0 + h) * 4
                             : n * OHW * (PADC ? g.mvalid : g.M) + (8 * mt0 + h) * OHW4 + int64_t(prem) * 4;
  const int oplanes = g.mvalid >> 2;  // PADC: output planes that exist
  float *yp = Y + yoff;
  const float *rp = residual ? residual + yoff : nullptr;
  const f32x4 *bq = bias ? reinterpret_cast<const f32x4 *>(bias + 32 * mt0 + 4 * h) : nullptr;
  // bias quads + residual quads of tile t+1 are requested before tile t is finished and stored, and the
  // activation is resolved once (dispatch_act), so the epilogue is two memory latencies, not one per element
  auto fetch = [&](f32x4(&bv)[4], f32x4(&rv)[4], int t) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      bv[q] = bq ? bq[8 * t + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
      const bool there = !PADC || 8 * (mt0 + t) + 2 * q + h < oplanes;
      rv[q] = (rp && there) ? *reinterpret_cast<const f32x4 *>(rp + (8 * t + 2 * q) * OHW4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  dispatch_act(act.kind, [&](auto kind_tag) {
    constexpr int KIND = decltype(kind_tag)::value;
    f32x4 bv[2][4], rv[2][4];
    fetch(bv[0], rv[0], 0);
#ifdef INFERA_CONV_PROBES
    if constexpr (PROBE == 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tc = __builtin_readcyclecounter();
    }
#endif
#pragma unroll
    for (int t = 0; t < MT; t++) {
      if (t + 1 < MT) fetch(bv[(t + 1) & 1], rv[(t + 1) & 1], t + 1);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = apply_act_c<KIND>((acc[t][4 * q + j] + bv[t & 1][q][j]) + rv[t & 1][q][j], act.a, act.b);
        if constexpr (DENSE) {
          const int f0 = 32 * (mt0 + t) + 8 * q + 4 * h;  // first of this quad's four features
          float *dst = yp + (8 * t + 2 * q) * 4;
          if (f0 + 3 < g.mvalid) {
            *reinterpret_cast<f32x4 *>(dst) = v;  // rows are mvalid floats apart: 4-byte aligned is all a dwordx4 store needs
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (f0 + j < g.mvalid) dst[j] = v[j];
          }
        } else if constexpr (PADC) {
          if (8 * (mt0 + t) + 2 * q + h < oplanes) *reinterpret_cast<f32x4 *>(yp + (8 * t + 2 * q) * OHW4) = v;
        } else {
          *reinterpret_cast<f32x4 *>(yp + (8 * t + 2 * q) * OHW4) = v;
        }
      }
    }
  });
#ifdef INFERA_CONV_PROBES
  if constexpr (PROBE == 5) {
    t3 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t4 = __builtin_readcyclecounter();
    if (lane == 0) {
      unsigned long long *st = g_conv_stamps[MT == 4];
      atomicAdd(st + 0, ta - t0);
      atomicAdd(st + 1, tb - ta);
      atomicAdd(st + 2, t1 - tb);
      atomicAdd(st + 3, t2 - t1);
      atomicAdd(st + 4, tc - t2);
      atomicAdd(st + 5, t3 - tc);
      atomicAdd(st + 6, t4 - t3);
      atomicAdd(st + 7, 1ull);
      atomicAdd(st + 8, w_vm);
      atomicAdd(st + 9, w_ring);
      atomicAdd(st + 10, w_bar);
    }
  }
#endif
}

// ---- weight-stationary persistent kernel --------------------------------------------------------------------------
// For convolutions whose packed weights (of one M-slice of 32*MT features) fit in LDS -- ResNet's 64-channel 3x3 layers
// (9*64*64*4 = 144 KB), the 64-channel stride-2 entry of layer2 in two 64-feature slices, every 1x1 downsample -- the tiled
// kernel's per-stage weight slab, its barrier per stage and its one-tile-per-workgroup prologue / epilogue are all
// overhead.  Here the structure is the fused MLP's (mlp_device.inc): a PERSISTENT workgroup (one per CU, NW waves) loads
// its weight slice into LDS once, then every wave walks 32-pixel tiles on its own -- no barrier after the prologue.  A
// stage is S 32-channel chunks of one tap: its A fragments come from LDS through the same unit pipeline (ds_read_b128
// ring + 4 MFMAs, sched_barrier-pinned), its B operands are gathered from the channel-quad planes one stage ahead.  The
// stage stream is CONTINUOUS across tiles: while the last stage of a tile runs, the first stage of the wave's next tile
// is already being gathered, and the epilogue's loads / stores drain under the next tile's MFMAs.
// Same packed weights as the tiled kernel (chunk-major, conv2d_tiled_pack), same summation order -> bit-identical results.
// (Measured and dropped: 12 waves per workgroup = 3 per SIMD for the 64-feature form -- needs 168 registers, spills 114,
// 1.96-2.11 ms against 1.73-1.78; 32-feature slices for 256-channel layers do not fit.  An A-fragment ring that runs on through
// stage and tile boundaries -- no P reads + wait at the head of a stage -- measured the same: the SIMD's other wave covers it.)
template <int MT, int S, int NW>
__global__ __launch_bounds__(NW * 64) void conv2d_ws_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                           const float *__restrict__ bias, const float *__restrict__ residual,
                                                           float *__restrict__ Y, int64_t total_pix, ConvGeom g, ActParam act) {
  constexpr int NB = 4 * S;   // B fragments (16 B per lane) per stage
  constexpr int U = NB * MT;  // units per stage
  constexpr int P = 3;        // A-fragment ring depth
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int OHW = g.OH * g.OW;
  const int MTtot = g.M / 32, mt0 = blockIdx.y * MT;
  const int CS = g.C / (32 * S), ntaps = g.kh * g.kw, nstages = ntaps * CS, nchunks = nstages * S;
  // ---- the slice's weights: chunk c of the packed blob holds MTtot KB-tiles; ours are [mt0, mt0 + MT) ----
  for (int i = threadIdx.x; i < nchunks * MT * 256; i += NW * 64) {
    const int chunk = i / (MT * 256), rem = i - chunk * (MT * 256);
    reinterpret_cast<f32x4 *>(wlds)[i] = reinterpret_cast<const f32x4 *>(Wp + (int64_t(chunk) * MTtot + mt0) * 1024)[rem];
  }
  __syncthreads();

  const int64_t ntiles = (total_pix + 31) >> 5;
  const int64_t tstride = int64_t(gridDim.x) * NW;
  int64_t tile = int64_t(blockIdx.x) * NW + wave;
  if (tile >= ntiles) return;
  const int HW4 = g.H * g.W * 4;  // floats per channel-quad plane
  const float *zp = g_zero_page + 4 * h;

  // ---- prefetch cursor: the (tile, stage) whose B operands are gathered next ----
  const float *p_xc = zp;  // receptive-field corner of this lane's pixel in the cursor's tile
  uint64_t p_ok = 0;       // taps of that pixel that lie inside the image
  int p_tap = 0, p_kx = 0, p_off = 0, p_base = 0;
  auto enter_tile = [&](int64_t t) {
    const int64_t pix = (t << 5) + r;
    const bool pvalid = t < ntiles && pix < total_pix;
    // (32-bit: the launcher keeps total_pix below 2^31 -- as 64-bit divisions this and the epilogue's were ~400 VALU instructions per
    // tile, squeezed into one unit's MFMA shadow)
    const unsigned pix32 = pvalid ? unsigned(pix) : 0u, n32 = pix32 / unsigned(OHW);
    const int64_t n = n32;
    const int prem = int(pix32 - n32 * unsigned(OHW));
    const int oh = int(unsigned(prem) / unsigned(g.OW)), ow = prem - oh * g.OW;
    const int ih0 = oh * g.sh - g.pt, iw0 = ow * g.sw - g.pl;
    p_xc = X + n * int64_t(g.H) * g.W * g.C + int64_t(h) * HW4 + (int64_t(ih0) * g.W + iw0) * 4;
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
  // gathers the cursor's stage into b, then advances the cursor (wrapping into the wave's next tile)
  auto gather = [&](f32x4(&b)[NB]) {
    const bool ok = (p_ok >> p_tap) & 1;
    const float *p = ok ? p_xc + p_off : zp;
    const int64_t pstride = ok ? 2 * int64_t(HW4) : 0;  // group q+1 = two channel-quad planes further
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
      if (p_base == CS * 2 * NB * HW4) {  // all channel blocks done: on to this wave's next tile (or a tile of zeros)
        p_tile += tstride;
        enter_tile(p_tile);
      }
    }
  };

  f32x16 acc[MT];
  auto zero_acc = [&] {
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
  };
  auto step = [&](const f32x4(&bc)[NB], f32x4(&bn)[NB], int stage) {
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(wlds + int64_t(stage) * S * MT * 1024) + lane;
    auto fidx = [](int u) { return (((u / MT) / 4 * MT + u % MT) * 4 + (u / MT) % 4) * 64; };
    f32x4 ring[P];
#pragma unroll
    for (int u = 0; u < P && u < U; u++) ring[u] = wl[fidx(u)];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int q = u / MT, t = u % MT;
      const f32x4 a = ring[u % P];
      if (u + P < U) ring[u % P] = wl[fidx(u + P)];
      if (u == 0) gather(bn);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bc[q][j], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // this lane's bias quads stay in registers for the workgroup's lifetime: a tile's epilogue has no load to wait for except
  // the residual (MT <= 2: 32 registers; wider tiles keep loading them, they have the registers' worth of accumulators)
  constexpr bool BRES = MT <= 2;
  const f32x4 *bq = bias ? reinterpret_cast<const f32x4 *>(bias + 32 * mt0 + 4 * h) : nullptr;
  f32x4 bres[BRES ? MT : 1][4];
  if constexpr (BRES) {
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int q = 0; q < 4; q++) bres[t][q] = bq ? bq[8 * t + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t OHW4 = int64_t(OHW) * 4;
  // (Tried: output offset and residual quads fetched when the tile STARTS, so that the epilogue has no memory latency in it --
  // 32 more live registers, 0.15-0.3 ms slower per ResNet-18 pass.)
  auto epilogue = [&](int64_t t) {
    const int64_t pix = (t << 5) + r;
    if (pix >= total_pix) return;
    const unsigned n32 = unsigned(pix) / unsigned(OHW);
    const int64_t n = n32;
    const int prem = int(unsigned(pix) - n32 * unsigned(OHW));
    const int64_t yoff = n * OHW * int64_t(g.M) + int64_t(8 * mt0 + h) * OHW4 + int64_t(prem) * 4;
    float *yp = Y + yoff;
    const float *rp = residual ? residual + yoff : nullptr;
    auto fetch = [&](f32x4(&bv)[4], f32x4(&rv)[4], int t) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if constexpr (BRES) bv[q] = bres[t][q];
        else bv[q] = bq ? bq[8 * t + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
        rv[q] = rp ? *reinterpret_cast<const f32x4 *>(rp + (8 * t + 2 * q) * OHW4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    dispatch_act(act.kind, [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
      f32x4 bv[2][4], rv[2][4];
      fetch(bv[0], rv[0], 0);
#pragma unroll
      for (int t = 0; t < MT; t++) {
        if (t + 1 < MT) fetch(bv[(t + 1) & 1], rv[(t + 1) & 1], t + 1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = apply_act_c<KIND>((acc[t][4 * q + j] + bv[t & 1][q][j]) + rv[t & 1][q][j], act.a, act.b);
          *reinterpret_cast<f32x4 *>(yp + (8 * t + 2 * q) * OHW4) = v;
        }
      }
    });
  };

  // One tile: its first stage is already in `ba`; on return the first stage of the wave's next tile is in `ba` when the
  // stage count is even and in `bb` when it is odd (the caller alternates the roles).
  auto run_tile = [&](f32x4(&ba)[NB], f32x4(&bb)[NB], int64_t t) {
    zero_acc();
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

// ---- patch kernel: few input channels, NCHW input (the network's first convolution) -------------------------
// With C = 3 a "channel quad" gather has nothing to load 16 bytes of, and the generic kernel's scalar gathers
// + bounds tests leave the matrix cores at ~30 %.  Here a persistent workgroup owns a (4*TR) x TC tile of
// output pixels, stages the tile's whole receptive field (all C channels, zero-padded) in LDS once --
// coalesced row reads of the caller's NCHW blob, double-buffered across tiles -- and every B operand is a
// conflict-free ds_read_b32: patch columns are de-interleaved by (col mod stride) so the wave's TC pixels
// read consecutive words.  The fragment-major weights (K padded to 8) and the per-k patch offsets stay in
// LDS for the whole launch.  Output: channel-quad planes, bias + activation fused.
struct PatchGeom {
  int TR, TC;       // per-wave tile: TR rows x TC columns of output pixels (TR * TC == 32); 4 waves stack vertically
  int PR, PC;       // patch rows / columns (input pixels) per workgroup tile
  int HALF, ROWS;   // de-interleaved column count per stride phase; words per patch row (padded)
  int PLANE;        // words per channel = PR * ROWS
  int K8;           // 8-wide k groups (C*kh*kw rounded up)
  int tiles_x, tiles_y;
  int NE;           // patch words each thread moves per tile
  int PP, RO;       // conv2d_stem_split6_kernel only (stem_split6_geom): words per PAIR of patch rows, offset of the pair's odd row
};
constexpr int kPatchMaxE = 16;

// The stem + max-pool kernel (POOL): a 512-thread workgroup owns an 8 x 7 tile of POOLED pixels, i.e. the 17 x 15 convolution
// pixels its 3x3/2 windows cover (255 = eight 32-pixel MFMA tiles, one per wave, pixel p -> row p/15, column p%15).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kPoolBlock = 512, kPoolCR = 17, kPoolCC = 15, kPoolTR = 8, kPoolTC = 7;
static PatchGeom patch_pool_geom(const ConvGeom &g, const PoolTail &pool) {
  PatchGeom p{};
  p.TC = kPoolCC;
  p.TR = 0;
  p.PR = (kPoolCR - 1) * g.sh + (g.kh - 1) * g.dh + 1;
  p.PC = (kPoolCC - 1) * g.sw + (g.kw - 1) * g.dw + 1;
  p.HALF = (p.PC + g.sw - 1) / g.sw;
  p.ROWS = g.sw * p.HALF;
  // consecutive convolution rows (15 pixels each) should sit 16 banks apart: a wave's 32 pixels span 2-3 rows
  for (int pad = 0; pad < 32; pad++)
    if (((p.ROWS + pad) * g.sh) % 32 == 16) {
      p.ROWS += pad;
      break;
    }
  p.PLANE = p.PR * p.ROWS;
  p.K8 = (g.C * g.kh * g.kw + 7) / 8;
  p.tiles_x = (pool.OW + kPoolTC - 1) / kPoolTC;
  p.tiles_y = (pool.OH + kPoolTR - 1) / kPoolTR;
  p.NE = (g.C * p.PR * p.PC + kPoolBlock - 1) / kPoolBlock;
  return p;
}

static size_t patch_pool_lds_bytes(const ConvGeom &g, const PatchGeom &p) {
  return (size_t(p.K8) * (g.M / 32) * 256 + size_t(p.K8) * 8 + 2 * size_t(g.C) * p.PLANE + 8 + 256 * size_t(g.M + 4)) * sizeof(float);
}

static PatchGeom patch_geom(const ConvGeom &g) {
  PatchGeom p{};
  p.TC = g.OW % 32 == 0 ? 32 : (g.OW % 16 == 0 ? 16 : (g.OW >= 24 ? 32 : (g.OW >= 12 ? 16 : 8)));
  p.TR = 32 / p.TC;
  p.PR = (4 * p.TR - 1) * g.sh + (g.kh - 1) * g.dh + 1;
  p.PC = (p.TC - 1) * g.sw + (g.kw - 1) * g.dw + 1;
  p.HALF = (p.PC + g.sw - 1) / g.sw;
  p.ROWS = g.sw * p.HALF;
  // the wave's second pixel row should start TC banks after the first: sh * ROWS == TC (mod 32)
  if (p.TR > 1)
    for (int pad = 0; pad < 32; pad++)
      if (((p.ROWS + pad) * g.sh) % 32 == p.TC % 32) {
        p.ROWS += pad;
        break;
      }
  p.PLANE = p.PR * p.ROWS;
  p.K8 = (g.C * g.kh * g.kw + 7) / 8;
  p.tiles_x = (g.OW + p.TC - 1) / p.TC;
  p.tiles_y = (g.OH + 4 * p.TR - 1) / (4 * p.TR);
  p.NE = (g.C * p.PR * p.PC + kBlock - 1) / kBlock;
  return p;
}

static size_t patch_lds_bytes(const ConvGeom &g, const PatchGeom &p) {
  return (size_t(p.K8) * (g.M / 32) * 256 + size_t(p.K8) * 8 + 2 * size_t(g.C) * p.PLANE + 4) * sizeof(float);
}

// K8C > 0: the number of k groups is a compile-time constant and the group loop is fully unrolled (register
// double-buffers indexed by constants: no rotation copies, no loop branches between MFMA batches -- the
// run-time loop leaves ~30 non-MFMA instructions per 8 MFMAs).  K8C == 0: any K8 (run-time loop).
//
// POOL: the convolution is followed by MaxPool 3x3 / stride 2 (pads 0 or 1) and the convolution's own output is never stored.
// Eight waves compute the 17 x 15 convolution pixels under an 8 x 7 tile of pooled pixels (bias + activation applied; pixels
// outside the convolution's output are written as -inf = the pooling's padding), park them in an LDS exchange tile
// [256 pixels][M + 4 floats] (17-quad pixel stride: the accumulator quads of 32 consecutive pixels hit 32 distinct bank quads)
// and every thread then takes the maximum of nine quads per pooled (pixel, channel quad) and stores it.  The row / column
// between neighbouring tiles is computed by both (256 / 224 = 1.14x the MFMA work) -- against a 3.3 GB tensor written, read back
// and a separate launch.  Two barriers per tile; the second one (exchange tile free again) is reached long after the last reader.
// (Tried on top of this: the exchange write of tile k-1 -- from a saved copy of its accumulators -- and the patch store of tile k+1
// moved into tile k's k loop as well, one quad / one word per unit, a barrier in the middle of the loop; nothing left between two
// loops but 32 register moves and one barrier.  2.75 ms against 2.60: the extra LDS traffic in the loop's in-order lgkmcnt stream
// and the second rendezvous cost more than the lock-step phases they replaced.)
template <int MT, int K8C, bool POOL = false>
__global__ __launch_bounds__(POOL ? kPoolBlock : kBlock) void conv2d_patch_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                             const float *__restrict__ bias, float *__restrict__ Y,
                                                             int64_t ntiles, ConvGeom g, PatchGeom pg, ActParam act, PoolTail pool) {
  constexpr int BS = POOL ? kPoolBlock : kBlock;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *wl = smem;                                                  // [K8][MT][64 lanes][4]
  int *ktab = reinterpret_cast<int *>(smem + pg.K8 * MT * 256);       // [K8][h][4] patch offsets of k = 8g + 4h + j
  float *patch = smem + pg.K8 * MT * 256 + pg.K8 * 8;                 // [2][C * PLANE]  (POOL: then the exchange tile)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int psz = g.C * pg.PLANE;

  // weights + offsets: once per (persistent) workgroup
  {
    const int nw4 = (pg.K8 * MT * 256 + pg.K8 * 8) / 4;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(Wp);
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    for (int i = threadIdx.x; i < nw4; i += BS) dst[i] = src[i];
  }
  // this thread's share of the patch: word e of the (c, row, col) enumeration -> global offset relative to
  // the patch corner, (row, col) for the bounds test, and the de-interleaved LDS slot
  int e_rel[kPatchMaxE], e_rc[kPatchMaxE], e_lds[kPatchMaxE];
#pragma unroll
  for (int i = 0; i < kPatchMaxE; i++) {
    const int e = threadIdx.x + i * BS;
    const int c = e / (pg.PR * pg.PC), rem = e - c * (pg.PR * pg.PC), row = rem / pg.PC, col = rem - row * pg.PC;
    const bool live = i < pg.NE && c < g.C;
    e_rel[i] = (c * g.H + row) * g.W + col;
    e_rc[i] = live ? (row << 16) | col : -1;
    e_lds[i] = live ? c * pg.PLANE + row * pg.ROWS + (col % g.sw) * pg.HALF + col / g.sw : -1;
  }
  // The staging code carries no branch: a word outside the image (or past this thread's share) loads the image's first
  // pixel and is zeroed by a select, a dead slot stores into four spare floats behind the two patch buffers.  (Guarded
  // element by element, each of the 16 slots was its own exec-masked block: ~400 scalar-heavy instructions per tile.)
  // Slots 8.. and 12.. are skipped by one uniform branch each when the share is that short.
  const bool ne8 = pg.NE <= 8, ne12 = pg.NE <= 12;
  const int tiles_per_img = pg.tiles_x * pg.tiles_y;
  // (32-bit: the launcher keeps ntiles below 2^31.  As 64-bit divisions, executed by every lane for every tile, the two
  // calls per tile were ~300 instructions: a tenth of a tile's time, outside the MFMA loop)
  auto tile_origin = [&](int64_t t, int &img, int &oy0, int &ox0) {
    const unsigned u = unsigned(t), im = u / unsigned(tiles_per_img), rem = u - im * unsigned(tiles_per_img);
    const unsigned ty = rem / unsigned(pg.tiles_x), tx = rem - ty * unsigned(pg.tiles_x);
    img = int(im);
    if constexpr (POOL) {  // first convolution pixel under the tile's first pooled pixel
      oy0 = int(ty) * kPoolTR * 2 - pool.pt;
      ox0 = int(tx) * kPoolTC * 2 - pool.pl;
    } else {
      oy0 = int(ty) * 4 * pg.TR;
      ox0 = int(tx) * pg.TC;
    }
  };
  // one slot of the patch of the tile at (img, oy0, ox0): a dead slot has row -1 / column 65535 -- never inside
  auto load_slot = [&](int i, const float *image, int iy0, int ix0) -> float {
    const int iy = iy0 + (e_rc[i] >> 16), ix = ix0 + (e_rc[i] & 0xffff);
    const bool ok = unsigned(iy) < unsigned(g.H) && unsigned(ix) < unsigned(g.W);
    const float x = image[ok ? iy0 * g.W + ix0 + e_rel[i] : 0];
    return ok ? x : 0.f;
  };
  auto load_patch = [&](float(&v)[kPatchMaxE], int img, int oy0, int ox0) {
    const int iy0 = oy0 * g.sh - g.pt, ix0 = ox0 * g.sw - g.pl;
    const float *image = X + int64_t(img) * g.C * g.H * g.W;
#pragma unroll
    for (int i = 0; i < kPatchMaxE; i++) {
      if ((i == 8 && ne8) || (i == 12 && ne12)) break;
      v[i] = load_slot(i, image, iy0, ix0);
    }
  };
  auto store_patch = [&](const float(&v)[kPatchMaxE], int buf, bool all) {  // all: every slot was fetched (dead ones too)
#pragma unroll
    for (int i = 0; i < kPatchMaxE; i++) {
      if (!all && ((i == 8 && ne8) || (i == 12 && ne12))) break;
      patch[e_lds[i] >= 0 ? buf * psz + e_lds[i] : 2 * psz + (i & 3)] = v[i];
    }
  };

  // lane's pixel inside the workgroup tile, and its word offset inside a patch channel
  // (POOL: pixel 255 of the 17 x 15 tile does not exist; its lanes recompute pixel 254 into an exchange slot nobody reads)
  const int pix = POOL ? min(wave * 32 + r, kPoolCR * kPoolCC - 1) : 0;
  const int py = POOL ? pix / kPoolCC : wave * pg.TR + r / pg.TC, px = POOL ? pix % kPoolCC : r % pg.TC;
  const int lbase = py * g.sh * pg.ROWS + px;
  const f32x4 *wfrag = reinterpret_cast<const f32x4 *>(wl) + lane;
  const int4 *ktab4 = reinterpret_cast<const int4 *>(ktab) + h;
  const int OHW = g.OH * g.OW;

  // the lane's bias quads: constant for the (persistent) workgroup's lifetime -- resident in registers, so a tile's epilogue
  // is add -> activation -> store with no load to wait for (the tiles are only 152 MFMAs long)
  f32x4 bres[MT][4];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int q = 0; q < 4; q++) bres[t][q] = bias ? reinterpret_cast<const f32x4 *>(bias)[h + 8 * t + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
  float pv[kPatchMaxE];
  // XCD x (= blockIdx.x % 8) owns the contiguous tile range [x*chunk, (x+1)*chunk): neighbouring tiles overlap in
  // their receptive fields, and the overlap should be found in THAT XCD's L2 (with a plain grid stride the
  // neighbours sat on eight different L2s and the blob was fetched from HBM 3.2 times)
  const int64_t chunk = (ntiles + 7) >> 3, t_end = min(ntiles, (int64_t(blockIdx.x & 7) + 1) * chunk);
  const int64_t tstep = (gridDim.x + 7 - (blockIdx.x & 7)) >> 3;  // workgroups on this XCD
  int64_t tile = int64_t(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  int img_n = 0, oy0_n = 0, ox0_n = 0;  // origin of the tile whose patch is being fetched (the next one; first: this one)
  if (tile < t_end) {
    tile_origin(tile, img_n, oy0_n, ox0_n);
    load_patch(pv, img_n, oy0_n, ox0_n);
    store_patch(pv, 0, false);
  }
  __syncthreads();
  // the first two offset quads and the first A fragments are the same for every tile: read once (a tile then starts with
  // its B words, not with an offset read -> B read -> MFMA chain of two LDS latencies)
  const int4 ko_first = ktab4[0], ko_second = ktab4[K8C > 1 ? 2 : 0];
  f32x4 a_first[MT];
#pragma unroll
  for (int t = 0; t < MT; t++) a_first[t] = wfrag[t * 64];
  // ---- POOL: the exchange tile and this thread's (at most two) pooled (pixel, channel quad) items ----
  constexpr int XQ = 8 * MT + 1, NQ = 8 * MT, PT = kPoolTR * kPoolTC;  // quads per pixel in the exchange tile; quads; pooled pixels
  // SHADOW: the pooling of tile k-1 runs inside tile k's MFMA loop (compile-time k loops of at least 22 units), one LDS read or
  // one quad of max per unit in the shadow of the unit's first MFMA, results stored through a buffer descriptor (lanes whose
  // pooled pixel lies outside the tensor carry offset -1 and are dropped by the bounds check: no branch in the loop; the first
  // tile's "previous tile" has a descriptor of zero bytes).  The exchange tile stays valid until the barrier in front of the
  // next exchange write.  Otherwise (short or run-time k loops) every thread pools right after the exchange, in lock step.
  constexpr bool SHADOW = POOL && K8C >= 6;
  f32x4 *exch = reinterpret_cast<f32x4 *>(patch + ((2 * psz + 4 + 3) & ~3));  // behind the patch buffers and their four spare floats
  int pwin[2] = {0, 0}, poff[2] = {-1, -1}, ppr[2] = {0, 0}, ppc[2] = {0, 0};
  if constexpr (POOL) {
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const int it = threadIdx.x + n * BS;
      const bool ok = it < NQ * PT;
      const int cq = ok ? it / PT : 0, pp = ok ? it % PT : 0, pr = pp / kPoolTC, pc = pp % kPoolTC;
      pwin[n] = ((2 * pr) * kPoolCC + 2 * pc) * XQ + cq;
      ppr[n] = pr;
      ppc[n] = pc;
      poff[n] = ok ? ((cq * pool.OH + pr) * pool.OW + pc) * 16 : -1;  // bytes from the image's pooled pixel (0, 0) of quad 0
    }
  }
  const unsigned pooled_img_bytes = unsigned(NQ) * unsigned(pool.OH) * unsigned(pool.OW) * 16u;
  auto pooled_rsrc = [&](int img, bool live) {
    const char *base = reinterpret_cast<const char *>(Y) + int64_t(__builtin_amdgcn_readfirstlane(img)) * pooled_img_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, live ? int(pooled_img_bytes) : 0, 0x00020000);
  };
  auto pooled_voff = [&](int n, int pr0, int pc0) {
    return (poff[n] >= 0 && pr0 + ppr[n] < pool.OH && pc0 + ppc[n] < pool.OW) ? poff[n] + (pr0 * pool.OW + pc0) * 16 : -1;
  };
  auto pooled_store = [&](const f32x4 &m, __amdgpu_buffer_rsrc_t rs, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, m), rs, voff, 0, 0);
  };
  auto pool_tile = [&](int img, int pr0, int pc0) {  // lock step: this thread's items of the tile in the exchange tile
    const __amdgpu_buffer_rsrc_t rs = pooled_rsrc(img, true);
#pragma unroll
    for (int n = 0; n < 2; n++) {
      if (n * BS >= NQ * PT) break;
      const f32x4 *win = exch + pwin[n];
      f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const f32x4 v = win[(i * kPoolCC + j) * XQ];
#pragma unroll
          for (int e = 0; e < 4; e++) m[e] = fmaxf(m[e], v[e]);
        }
      pooled_store(m, rs, pooled_voff(n, pr0, pc0));
    }
  };
  int img_p = 0, pr0_p = 0, pc0_p = 0;  // the tile whose exchange tile is waiting to be pooled
  bool have_p = false;
  int buf = 0;
  for (; tile < t_end; tile += tstep, buf ^= 1) {
    const int64_t next = tile + tstep;
    const int img = img_n, oy0 = oy0_n, ox0 = ox0_n;
    // (SHADOW) descriptor and offsets of the previous tile's pooled outputs, and the in-loop pooling state
    const __amdgpu_buffer_rsrc_t rs_p = pooled_rsrc(img_p, have_p);
    const int voff_p[2] = {pooled_voff(0, pr0_p, pc0_p), pooled_voff(1, pr0_p, pc0_p)};
    f32x4 ptmp[2], pm;
    auto shadow_pool = [&](int uu) {  // item 0: units 0..10, item 1: units 11..21 (nine reads, nine maxima one unit later, the store)
      const int n = uu / 11, st = uu - 11 * n;
      if (n > 1) return;
      if (st < 9) ptmp[st & 1] = exch[pwin[n] + ((st / 3) * kPoolCC + st % 3) * XQ];
      if (st == 0) pm = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // (as the pooling kernels: a window of NaNs only pools to -inf)
      if (st >= 1 && st <= 9) {
#pragma unroll
        for (int e = 0; e < 4; e++) pm[e] = fmaxf(pm[e], ptmp[(st - 1) & 1][e]);
      }
      if (st == 10) pooled_store(pm, rs_p, voff_p[n]);
    };
    // The next tile's patch is fetched under this tile's MFMAs.  Compile-time k loop: slot by slot INSIDE the loop, each slot's
    // bounds test, address and load in the shadow of one MFMA (ahead of the loop the ~200 instructions were exposed: every wave of
    // the workgroup is in the same phase).  The last tile re-fetches itself (no branch in the loop; nothing is stored).
    tile_origin(next < t_end ? next : tile, img_n, oy0_n, ox0_n);
    const int iy0_n = oy0_n * g.sh - g.pt, ix0_n = ox0_n * g.sw - g.pl;
    const float *image_n = X + int64_t(img_n) * g.C * g.H * g.W;
    if constexpr (K8C == 0) load_patch(pv, img_n, oy0_n, ox0_n);

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    const float *pb = patch + buf * psz + lbase;
    // software pipeline: offsets two groups ahead, B words and A fragments one group ahead
    if constexpr (K8C > 0) {
      int4 ko[3];
      float b[2][4];
      f32x4 a[2][MT];
      ko[0] = ko_first;
      ko[1] = ko_second;
#pragma unroll
      for (int j = 0; j < 4; j++) b[0][j] = pb[(&ko[0].x)[j]];
#pragma unroll
      for (int t = 0; t < MT; t++) a[0][t] = a_first[t];
#pragma unroll
      for (int grp = 0; grp < K8C; grp++) {
        // A group = four units of (one B word x MT MFMAs).  Each unit also ISSUES a quarter of the next group's LDS reads -- its
        // B word j, A fragment j, the offsets of the group after that -- in the shadow of the unit's first MFMA: a whole group
        // (8 MFMAs at MT = 2) before they are used, and with the matrix pipe already busy.  (Left to the scheduler the reads sank
        // behind the group's last MFMA and every group opened with s_waitcnt lgkmcnt(0); pinned in front of the group's first MFMA
        // the pipe idled while a dozen reads and address adds issued.)
        const int cur = grp & 1, nxt = cur ^ 1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (grp + 1 < K8C) {
            b[nxt][j] = pb[(&ko[(grp + 1) % 3].x)[j]];
            if (j < MT) a[nxt][j] = wfrag[((grp + 1) * MT + j) * 64];
          }
          if (j == 0 && grp + 2 < K8C) ko[(grp + 2) % 3] = ktab4[(grp + 2) * 2];  // (used from the next group's first unit on)
          // patch slots of the next tile, spread evenly over the 4 * K8C units
#pragma unroll
          for (int sl = 0; sl < kPatchMaxE; sl++)
            if (sl * (4 * K8C) / kPatchMaxE == grp * 4 + j) pv[sl] = load_slot(sl, image_n, iy0_n, ix0_n);
          if constexpr (SHADOW) shadow_pool(grp * 4 + j);
#pragma unroll
          for (int t = 0; t < MT; t++)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][t][j], b[cur][j], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // the unit's first MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);  // address arithmetic of ...
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // ... the next group's reads (and one of the pooling's)
          __builtin_amdgcn_sched_group_barrier(0x020, K8C < 4 ? 4 : (K8C < 8 ? 2 : 1), 0);  // ... and of this unit's patch slot(s), if any
          __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // (a pooled store)
          if (MT > 1) __builtin_amdgcn_sched_group_barrier(0x008, MT - 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    int4 ko_n = ktab4[0];
    f32x4 a_n[MT];
    float b_n[4];
#pragma unroll
    for (int j = 0; j < 4; j++) b_n[j] = pb[(&ko_n.x)[j]];
#pragma unroll
    for (int t = 0; t < MT; t++) a_n[t] = wfrag[t * 64];
    ko_n = ktab4[pg.K8 > 1 ? 2 : 0];
    for (int grp = 0; grp < pg.K8; grp++) {
      f32x4 a[MT];
      float b[4];
#pragma unroll
      for (int t = 0; t < MT; t++) a[t] = a_n[t];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = b_n[j];
      if (grp + 1 < pg.K8) {
        const int4 ko = ko_n;
        if (grp + 2 < pg.K8) ko_n = ktab4[(grp + 2) * 2];
#pragma unroll
        for (int j = 0; j < 4; j++) b_n[j] = pb[(&ko.x)[j]];
#pragma unroll
        for (int t = 0; t < MT; t++) a_n[t] = wfrag[((grp + 1) * MT + t) * 64];
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[j], acc[t], 0, 0, 0);
    }

    }

    // epilogue: lane (r,h) holds pixel (oy, ox), channels 32t + 8q + 4h + j -> channel quad 8t + 2q + h
    const int oy = oy0 + py, ox = ox0 + px;
    if constexpr (POOL) {
      const bool inside = oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW;
      __syncthreads();  // the previous tile's pooling has read the exchange tile (SHADOW: inside the k loop everybody has just left)
      dispatch_act(act.kind, [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
        for (int t = 0; t < MT; t++) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = inside ? apply_act_c<KIND>(acc[t][4 * q + j] + bres[t][q][j], act.a, act.b) : -INFINITY;
            exch[(wave * 32 + r) * XQ + 8 * t + 2 * q + h] = v;
          }
        }
      });
      // (unconditional -- the last tile parks its own re-fetched patch: a fetch that is consumed on one path only leaves the loop
      // head with an s_waitcnt vmcnt that also drains the pooled stores issued just before it, every tile)
      store_patch(pv, buf ^ 1, K8C > 0);
      __syncthreads();
      img_p = img;
      pr0_p = (oy0 + pool.pt) >> 1;
      pc0_p = (ox0 + pool.pl) >> 1;
      have_p = true;
      if constexpr (!SHADOW) pool_tile(img_p, pr0_p, pc0_p);
      continue;  // (both barriers of this tile are behind us; the patch for the next tile is in place)
    }
    if (oy < g.OH && ox < g.OW) {
      // g.mvalid > 0: M was padded to whole 32-feature tiles (stems with 16 / 24 outputs); planes past mvalid/4 do not exist
      const int mreal = g.mvalid > 0 ? g.mvalid : g.M;
      float *yp = Y + int64_t(img) * OHW * mreal + (int64_t(h) * OHW + oy * g.OW + ox) * 4;
      dispatch_act(act.kind, [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
        for (int t = 0; t < MT; t++) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = apply_act_c<KIND>(acc[t][4 * q + j] + bres[t][q][j], act.a, act.b);
            if (4 * (8 * t + 2 * q + h) < mreal) *reinterpret_cast<f32x4 *>(yp + int64_t(8 * t + 2 * q) * OHW * 4) = v;
          }
        }
      });
    }
    store_patch(pv, buf ^ 1, K8C > 0);  // (unconditional: see the pooled epilogue)
    __syncthreads();
  }
  if constexpr (SHADOW)
    if (have_p) pool_tile(img_p, pr0_p, pc0_p);  // the workgroup's last tile has no k loop behind it to hide under
}

// ---- stem + max-pool, TWO half-channel workgroups per CU (round 3) ------------------------------------------------------
// The 512-thread kernel above keeps one 146 KB workgroup per CU: its eight waves leave the k loop together, write the
// exchange tile together, park the next patch together and pass two barriers together -- phases in which the matrix pipe idles
// (mfma_busy 0.70 over the launch, 0.90 inside the k loops).  Here a 64-feature stem runs as TWO independent 256-thread
// workgroups per CU, each computing 32 of the 64 features for the same tile sequence: wave w of a workgroup owns pixel tiles w
// and w + 4 of the 17 x 15 tile (two accumulator tiles, ONE weight fragment per unit shared by both), so a workgroup does the
// same 152 MFMAs per wave per tile as before; 76 KB of LDS each (its half of the weights, ONE patch buffer -- both patch stores
// of the pooled flow sit between the tile's two barriers, when nobody reads the patch -- and a [256][32 + 4] exchange tile).
// Max-pooling is per channel, so the halves never talk to each other; the odd half starts half a tile late, so one
// workgroup's exchange / patch / barrier phase falls into the other's k loop.  Same k order per output element as the other
// stem kernels -> bit-identical (tests/test_stem_pool_gpu.py).
constexpr int kPool2Block = 256;
static size_t patch_pool2_lds_bytes(const ConvGeom &g, const PatchGeom &p) {
  return (size_t(p.K8) * 256 + size_t(p.K8) * 8 + size_t(g.C) * p.PLANE + 8 + 256 * size_t(32 + 4)) * sizeof(float);
}

template <int K8C>
__global__ __launch_bounds__(kPool2Block, 2) void conv2d_stem_pool2_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                                          const float *__restrict__ bias, float *__restrict__ Y, int64_t ntiles,
                                                                          ConvGeom g, PatchGeom pg, ActParam act, PoolTail pool, int desync) {
  constexpr int BS = kPool2Block, PW = 2;
  static_assert(K8C >= 6, "the pooling runs in the shadow of a compile-time k loop of at least 22 units");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *wl = smem;                                     // [K8][64 lanes][4]: this half's 32 features
  int *ktab = reinterpret_cast<int *>(smem + K8C * 256);  // [K8][h][4] patch offsets of k = 8g + 4h + j
  float *patch = smem + K8C * 256 + K8C * 8;             // [C * PLANE] (+ 4 spare floats), then the exchange tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int psz = g.C * pg.PLANE;
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, half = wg & 1, pair = wg >> 1;

  // this half's weight fragments out of the [K8][2][64][4] blob, and the offsets behind it: once per (persistent) workgroup
  for (int i = threadIdx.x; i < K8C * 64; i += BS)
    reinterpret_cast<f32x4 *>(wl)[i] = reinterpret_cast<const f32x4 *>(Wp)[((i >> 6) * 2 + half) * 64 + (i & 63)];
  for (int i = threadIdx.x; i < K8C * 8; i += BS) ktab[i] = reinterpret_cast<const int *>(Wp)[K8C * 2 * 256 + i];

  int e_rel[kPatchMaxE], e_rc[kPatchMaxE], e_lds[kPatchMaxE];
#pragma unroll
  for (int i = 0; i < kPatchMaxE; i++) {
    const int e = threadIdx.x + i * BS;
    const int c = e / (pg.PR * pg.PC), rem = e - c * (pg.PR * pg.PC), row = rem / pg.PC, col = rem - row * pg.PC;
    const bool live = c < g.C;
    e_rel[i] = (c * g.H + row) * g.W + col;
    e_rc[i] = live ? (row << 16) | col : -1;
    e_lds[i] = live ? c * pg.PLANE + row * pg.ROWS + (col % g.sw) * pg.HALF + col / g.sw : -1;
  }
  const int tiles_per_img = pg.tiles_x * pg.tiles_y;
  auto tile_origin = [&](int64_t t, int &img, int &oy0, int &ox0) {
    const unsigned u = unsigned(t), im = u / unsigned(tiles_per_img), rem = u - im * unsigned(tiles_per_img);
    const unsigned ty = rem / unsigned(pg.tiles_x), tx = rem - ty * unsigned(pg.tiles_x);
    img = int(im);
    oy0 = int(ty) * kPoolTR * 2 - pool.pt;
    ox0 = int(tx) * kPoolTC * 2 - pool.pl;
  };
  auto load_slot = [&](int i, const float *image, int iy0, int ix0) -> float {
    const int iy = iy0 + (e_rc[i] >> 16), ix = ix0 + (e_rc[i] & 0xffff);
    const bool ok = unsigned(iy) < unsigned(g.H) && unsigned(ix) < unsigned(g.W);
    const float x = image[ok ? iy0 * g.W + ix0 + e_rel[i] : 0];
    return ok ? x : 0.f;
  };
  auto store_patch = [&](const float(&v)[kPatchMaxE]) {
#pragma unroll
    for (int i = 0; i < kPatchMaxE; i++) patch[e_lds[i] >= 0 ? e_lds[i] : psz + (i & 3)] = v[i];
  };

  int lbase[PW], py[PW], px[PW];
#pragma unroll
  for (int p = 0; p < PW; p++) {
    const int pix = min((wave + 4 * p) * 32 + r, kPoolCR * kPoolCC - 1);
    py[p] = pix / kPoolCC;
    px[p] = pix % kPoolCC;
    lbase[p] = py[p] * g.sh * pg.ROWS + px[p];
  }
  const f32x4 *wfrag = reinterpret_cast<const f32x4 *>(wl) + lane;
  const int4 *ktab4 = reinterpret_cast<const int4 *>(ktab) + h;
  f32x4 bres[4];
#pragma unroll
  for (int q = 0; q < 4; q++) bres[q] = bias ? reinterpret_cast<const f32x4 *>(bias)[h + 8 * half + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
  float pv[kPatchMaxE];
  // XCD x owns the contiguous tile range [x*chunk, (x+1)*chunk); the two halves of a pair walk the same tiles
  const int64_t chunk = (ntiles + 7) >> 3, t_end = min(ntiles, (int64_t(xcd) + 1) * chunk);
  const int64_t tstep = ((gridDim.x + 7 - xcd) >> 3) >> 1;  // workgroup PAIRS on this XCD (the launcher keeps that even)
  int64_t tile = int64_t(xcd) * chunk + pair;
  int img_n = 0, oy0_n = 0, ox0_n = 0;
  if (tile < t_end) {
    tile_origin(tile, img_n, oy0_n, ox0_n);
    const int iy0 = oy0_n * g.sh - g.pt, ix0 = ox0_n * g.sw - g.pl;
    const float *image = X + int64_t(img_n) * g.C * g.H * g.W;
#pragma unroll
    for (int i = 0; i < kPatchMaxE; i++) pv[i] = load_slot(i, image, iy0, ix0);
    store_patch(pv);
  }
  __syncthreads();
  if (half)
    for (int i = 0; i < desync; i++) __builtin_amdgcn_s_sleep(127);  // ~3.5 us each: half a tile behind the even half
  const int4 ko_first = ktab4[0], ko_second = ktab4[2];
  const f32x4 a_first = wfrag[0];
  constexpr int XQ = 9, NQ = 8, PT = kPoolTR * kPoolTC;  // quads per pixel in the exchange tile; this half's quads; pooled pixels
  f32x4 *exch = reinterpret_cast<f32x4 *>(patch + ((psz + 4 + 3) & ~3));
  int pwin[2] = {0, 0}, poff[2] = {-1, -1}, ppr[2] = {0, 0}, ppc[2] = {0, 0};
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const int it = threadIdx.x + n * BS;
    const bool ok = it < NQ * PT;
    const int cq = ok ? it / PT : 0, pp = ok ? it % PT : 0, pr = pp / kPoolTC, pc = pp % kPoolTC;
    pwin[n] = ((2 * pr) * kPoolCC + 2 * pc) * XQ + cq;
    ppr[n] = pr;
    ppc[n] = pc;
    poff[n] = ok ? (((cq + NQ * half) * pool.OH + pr) * pool.OW + pc) * 16 : -1;
  }
  const unsigned pooled_img_bytes = unsigned(g.M / 4) * unsigned(pool.OH) * unsigned(pool.OW) * 16u;
  auto pooled_rsrc = [&](int img, bool live) {
    const char *base = reinterpret_cast<const char *>(Y) + int64_t(__builtin_amdgcn_readfirstlane(img)) * pooled_img_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, live ? int(pooled_img_bytes) : 0, 0x00020000);
  };
  auto pooled_voff = [&](int n, int pr0, int pc0) {
    return (poff[n] >= 0 && pr0 + ppr[n] < pool.OH && pc0 + ppc[n] < pool.OW) ? poff[n] + (pr0 * pool.OW + pc0) * 16 : -1;
  };
  auto pooled_store = [&](const f32x4 &m, __amdgpu_buffer_rsrc_t rs, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, m), rs, voff, 0, 0);
  };
  int img_p = 0, pr0_p = 0, pc0_p = 0;
  bool have_p = false;
  for (; tile < t_end; tile += tstep) {
    const int64_t next = tile + tstep;
    const int oy0 = oy0_n, ox0 = ox0_n, img = img_n;
    const __amdgpu_buffer_rsrc_t rs_p = pooled_rsrc(img_p, have_p);
    const int voff_p[2] = {pooled_voff(0, pr0_p, pc0_p), pooled_voff(1, pr0_p, pc0_p)};
    f32x4 ptmp[2], pm;
    auto shadow_pool = [&](int uu) {  // item 0: units 0..10, item 1: units 11..21
      const int n = uu / 11, st = uu - 11 * n;
      if (n > 1) return;
      if (st < 9) ptmp[st & 1] = exch[pwin[n] + ((st / 3) * kPoolCC + st % 3) * XQ];
      if (st == 0) pm = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (st >= 1 && st <= 9) {
#pragma unroll
        for (int e = 0; e < 4; e++) pm[e] = fmaxf(pm[e], ptmp[(st - 1) & 1][e]);
      }
      if (st == 10) {
        pooled_store(pm, rs_p, voff_p[n]);
      }
    };
    tile_origin(next < t_end ? next : tile, img_n, oy0_n, ox0_n);
    const int iy0_n = oy0_n * g.sh - g.pt, ix0_n = ox0_n * g.sw - g.pl;
    const float *image_n = X + int64_t(img_n) * g.C * g.H * g.W;

    f32x16 acc[PW];
#pragma unroll
    for (int p = 0; p < PW; p++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[p][i] = 0.f;
    const float *pb0 = patch + lbase[0], *pb1 = patch + lbase[1];
    int4 ko[3];
    float b[2][PW][4];
    f32x4 a[2];
    ko[0] = ko_first;
    ko[1] = ko_second;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      b[0][0][j] = pb0[(&ko[0].x)[j]];
      b[0][1][j] = pb1[(&ko[0].x)[j]];
    }
    a[0] = a_first;
#pragma unroll
    for (int grp = 0; grp < K8C; grp++) {
      const int cur = grp & 1, nxt = cur ^ 1;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (grp + 1 < K8C) {
          b[nxt][0][j] = pb0[(&ko[(grp + 1) % 3].x)[j]];
          b[nxt][1][j] = pb1[(&ko[(grp + 1) % 3].x)[j]];
          if (j == 0) a[nxt] = wfrag[(grp + 1) * 64];
        }
        if (j == 0 && grp + 2 < K8C) ko[(grp + 2) % 3] = ktab4[(grp + 2) * 2];
#pragma unroll
        for (int sl = 0; sl < kPatchMaxE; sl++)
          if (sl * (4 * K8C) / kPatchMaxE == grp * 4 + j) pv[sl] = load_slot(sl, image_n, iy0_n, ix0_n);
        shadow_pool(grp * 4 + j);
#pragma unroll
        for (int p = 0; p < PW; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][j], b[cur][p][j], acc[p], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // the unit's first MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);  // address arithmetic of ...
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);   // ... the next group's reads (and one of the pooling's)
        __builtin_amdgcn_sched_group_barrier(0x020, K8C < 8 ? 2 : 1, 0);  // ... and of this unit's patch slot(s), if any
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // (a pooled store)
        __builtin_amdgcn_sched_group_barrier(0x008, PW - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // epilogue: lane (r, h) of pixel tile p holds pixel (oy0 + py[p], ox0 + px[p]), channels 32 half + 8q + 4h + j -> this half's quad 2q + h
    __syncthreads();  // the previous tile's pooling has read the exchange tile (inside the k loop everybody has just left)
    dispatch_act(act.kind, [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
      for (int p = 0; p < PW; p++) {
        const int oy = oy0 + py[p], ox = ox0 + px[p];
        const bool inside = oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = inside ? apply_act_c<KIND>(acc[p][4 * q + j] + bres[q][j], act.a, act.b) : -INFINITY;
          exch[((wave + 4 * p) * 32 + r) * XQ + 2 * q + h] = v;
        }
      }
    });
    store_patch(pv);  // (unconditional: the last tile parks its own re-fetched patch, see conv2d_patch_kernel)
    __syncthreads();
    img_p = img;
    pr0_p = (oy0 + pool.pt) >> 1;
    pc0_p = (ox0 + pool.pl) >> 1;
    have_p = true;
  }
  if (have_p) {  // the workgroup's last tile has no k loop behind it to hide under
    const __amdgpu_buffer_rsrc_t rs = pooled_rsrc(img_p, true);
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const f32x4 *win = exch + pwin[n];
      f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const f32x4 v = win[(i * kPoolCC + j) * XQ];
#pragma unroll
          for (int e = 0; e < 4; e++) m[e] = fmaxf(m[e], v[e]);
        }
      const int voff = pooled_voff(n, pr0_p, pc0_p);
      pooled_store(m, rs, voff);
    }
  }
}

constexpr int kStemKB = 11;
// ---- the stem + max-pool in the DEFAULT arithmetic: bf16 x three exact parts, six MFMAs per product (round 3) ----------------------
// conv2d_stem_pool2_kernel's tile flow (two half-channel workgroups per CU) with the inner product of conv2d_split6_kernel (conv_split.hip).
// K = C*kh*kw = 147 as eleven k-blocks of 16 (instead of 152 exact-fp32 k-steps of 64 cycles).  k order: a lane half's eight k-values of a
// k-block are ONE filter row (c, ky), kx in the order 0 2 4 6 1 3 5 7 (kx = 7: zero weight).  The patch rows are de-interleaved by column
// parity (stride 2), so those are four consecutive words of the row's even half and four of its odd half: one row offset per k-block (eleven
// registers for the kernel's lifetime, no offset table).  21 filter rows = 11 k-blocks (the 22nd row has zero weights).
// No scales, nothing to track, nothing about the input
// has to hold.  The patch is cut ONCE, when it is parked: a patch word is the pair {hi | mid << 16}, {lo} (8 bytes), so the k loop assembles its
// three B fragments with three v_perm_b32 per pair of words (first version: fp32 patch, cut in the k loop -- 44 VALU instructions per k-block and
// pixel tile: 2.24 ms against 1.9).  The weights are three bf16 fragments per k-block (33 KB for this half's 32 features), and to fit two
// workgroups per CU the exchange tile ALIASES the patch (37 KB each): the pooling runs
// between the k loops (four barriers per tile instead of two: out of the k loop -> exchange tile written -> pooled and stored -> next patch parked)
// rather than in the next tile's shadow.  Same tile flow otherwise (two half-channel workgroups per CU, the odd half half a tile late).
// The patch layout of this kernel (8-byte words).  A lane group of a patch-word read (ds_read_b64: 32 lanes = 32 consecutive pixels of the 15-wide
// pixel tile, two to four convolution rows) is conflict-free when pixel i sits on 8-byte slot i mod 32, i.e. when consecutive convolution rows are
// 15 slots apart (mod 32).  A convolution row is TWO patch rows down (stride 2), so no single row pitch can do that (twice a pitch is even -- the
// 16-slot pitch of the exact-fp32 stems left two lanes of every group on a taken slot: every read took twice its cycles, 0.63 M of the 0.87 M
// conflict cycles per CU of the round-4 PMC pass): patch rows are laid out in PAIRS, PP = 15 (mod 32) words per pair, the odd row RO words in.
static PatchGeom stem_split6_geom(const ConvGeom &g, const PoolTail &pool) {
  PatchGeom p = patch_pool_geom(g, pool);
  const int row_words = g.sw * p.HALF;
  p.RO = row_words;
  p.PP = 2 * row_words;
  while (p.PP % 32 != 15) p.PP++;
  p.ROWS = 0;  // (no uniform row pitch here)
  p.PLANE = (p.PR + 1) / 2 * p.PP;
  return p;
}
static size_t stem_split6_lds_bytes(const ConvGeom &g, const PatchGeom &p) {
  const size_t patch_b = (size_t(g.C) * p.PLANE + 8) * 8, exch_b = 256 * size_t(32 + 4) * 4;
  return size_t(kStemKB) * 3072 + std::max(patch_b, exch_b) + 64;
}
using bf16x8_s = __attribute__((ext_vector_type(8))) __bf16;
#ifdef INFERA_STEM_TIMING  // tools/ubench/stem_phases.hip: cycle sums per phase of the first workgroup pair (wave 0 of each half)
__device__ unsigned long long g_stem_phase[2][10];
#define STEM_T(i)                                                   \
  if (timing) {                                                     \
    const unsigned long long t_now = __builtin_readcyclecounter();  \
    tsum[i] += t_now - tlast;                                       \
    tlast = t_now;                                                  \
  }
#else
#define STEM_T(i)
#endif

template <int NW>  // waves per workgroup: 4 (two pixel tiles per wave) or 8 (one: four waves per SIMD with two workgroups per CU)
__global__ __launch_bounds__(NW * 64, 2) void conv2d_stem_split6_kernel(const float *__restrict__ X, const float *__restrict__ Wp,
                                                                           const float *__restrict__ bias, float *__restrict__ Y, int64_t ntiles,
                                                                           ConvGeom g, PatchGeom pg, ActParam act, PoolTail pool, int desync) {
  constexpr int BS = NW * 64, PW = 8 / NW, KBC = kStemKB;
  constexpr int PE = kPatchMaxE * kPool2Block / BS;  // patch words a thread fetches and parks per tile
  constexpr int NI = 2 * kPool2Block / BS;           // pooled (pixel, channel quad) items per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  u32x4_t *wl = reinterpret_cast<u32x4_t *>(smem);           // [KBC][hi, mid, lo][64 lanes]: this half's 32 features
  uint2 *patch = reinterpret_cast<uint2 *>(smem + KBC * 768);  // [C * PLANE] cut words {hi | mid << 16, lo} (+ 8 spare) ...
  f32x4 *exch = reinterpret_cast<f32x4 *>(patch);            // ... and, between the k loops, the exchange tile [256][9 quads]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int psz = g.C * pg.PLANE;
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, half = wg & 1, pair = wg >> 1;
  auto rowoff = [&](int row) { return (row >> 1) * pg.PP + (row & 1) * pg.RO; };  // (stem_split6_geom)

  // blob: [KBC][half][part][64][4 dwords]
  for (int i = threadIdx.x; i < KBC * 192; i += BS)
    wl[i] = reinterpret_cast<const u32x4_t *>(Wp)[(((i / 192) * 2 + half) * 3 + (i % 192) / 64) * 64 + (i & 63)];
  int soff[2 * KBC];
#pragma unroll
  for (int rr = 0; rr < 2 * KBC; rr++) {
    const int c = rr / g.kh, ky = rr - c * g.kh;
    soff[rr] = rr < g.C * g.kh ? c * pg.PLANE + rowoff(ky) : 0;
  }
  int e_rel[PE], e_rc[PE], e_lds[PE];
#pragma unroll
  for (int i = 0; i < PE; i++) {
    const int e = threadIdx.x + i * BS;
    const int c = e / (pg.PR * pg.PC), rem = e - c * (pg.PR * pg.PC), row = rem / pg.PC, col = rem - row * pg.PC;
    const bool live = c < g.C;
    e_rel[i] = (c * g.H + row) * g.W + col;
    e_rc[i] = live ? (row << 16) | col : -1;
    e_lds[i] = live ? c * pg.PLANE + rowoff(row) + (col % g.sw) * pg.HALF + col / g.sw : -1;
  }
  const int tiles_per_img = pg.tiles_x * pg.tiles_y;
  auto tile_origin = [&](int64_t t, int &img, int &oy0, int &ox0) {
    const unsigned u = unsigned(t), im = u / unsigned(tiles_per_img), rem = u - im * unsigned(tiles_per_img);
    const unsigned ty = rem / unsigned(pg.tiles_x), tx = rem - ty * unsigned(pg.tiles_x);
    img = int(im);
    oy0 = int(ty) * kPoolTR * 2 - pool.pt;
    ox0 = int(tx) * kPoolTC * 2 - pool.pl;
  };
  auto load_slot = [&](int i, const float *image, int iy0, int ix0) -> float {
    const int iy = iy0 + (e_rc[i] >> 16), ix = ix0 + (e_rc[i] & 0xffff);
    const bool ok = unsigned(iy) < unsigned(g.H) && unsigned(ix) < unsigned(g.W);
    const float x = image[ok ? iy0 * g.W + ix0 + e_rel[i] : 0];
    return ok ? x : 0.f;
  };
  // park a patch: first every word of the region nobody parks (the odd half's fourth word of the last pixels: kx = 7, weight zero -- and what
  // the exchange tile left there) is made finite again, then the fetched words
  auto park_patch = [&](const float(&v)[PE]) {
    // (one word per patch row is read but never parked -- the odd half's word HALF - 1 = column 2 HALF - 1 past the patch, kx = 7 of the last
    //  pixels, weight zero: it must be finite, and the exchange tile has been there)
    for (int i = threadIdx.x; i < g.C * pg.PR; i += BS) patch[(i / pg.PR) * pg.PLANE + rowoff(i % pg.PR) + 2 * pg.HALF - 1] = uint2{0u, 0u};
#pragma unroll
    for (int i = 0; i < PE; i++) {  // exact cut: hi = top 16 bits, mid = top 16 bits of the rest, lo = what is left (8 bits: exact in bf16)
      const unsigned x = __float_as_uint(v[i]);
      const float r1 = v[i] - __uint_as_float(x & 0xffff0000u);
      const unsigned y = __float_as_uint(r1);
      const float r2 = r1 - __uint_as_float(y & 0xffff0000u);
      patch[e_lds[i] >= 0 ? e_lds[i] : psz + (i & 3)] = uint2{__builtin_amdgcn_perm(y, x, 0x07060302u), __float_as_uint(r2) >> 16};
    }
  };

  int lbase[PW], py[PW], px[PW];
#pragma unroll
  for (int p = 0; p < PW; p++) {
    const int pix = min((wave + NW * p) * 32 + r, kPoolCR * kPoolCC - 1);
    py[p] = pix / kPoolCC;
    px[p] = pix % kPoolCC;
    lbase[p] = py[p] * pg.PP + px[p];  // (stride 2: convolution row py starts at patch row 2 py)
  }
  const u32x4_t *wfrag = wl + lane;
  f32x4 bres[4];
#pragma unroll
  for (int q = 0; q < 4; q++) bres[q] = bias ? reinterpret_cast<const f32x4 *>(bias)[h + 8 * half + 2 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
  float pv[PE];
  const int64_t chunk = (ntiles + 7) >> 3, t_end = min(ntiles, (int64_t(xcd) + 1) * chunk);
  const int64_t tstep = ((gridDim.x + 7 - xcd) >> 3) >> 1;
  int64_t tile = int64_t(xcd) * chunk + pair;
  int img_n = 0, oy0_n = 0, ox0_n = 0;
  if (tile < t_end) {
    tile_origin(tile, img_n, oy0_n, ox0_n);
    const int iy0 = oy0_n * g.sh - g.pt, ix0 = ox0_n * g.sw - g.pl;
    const float *image = X + int64_t(img_n) * g.C * g.H * g.W;
#pragma unroll
    for (int i = 0; i < PE; i++) pv[i] = load_slot(i, image, iy0, ix0);
  }
  __syncthreads();  // (the weights)
  if (tile < t_end) park_patch(pv);
  __syncthreads();
  if (half)
    for (int i = 0; i < desync; i++) __builtin_amdgcn_s_sleep(127);
  constexpr int XQ = 9, NQ = 8, PT = kPoolTR * kPoolTC;
  int pwin[NI], poff[NI], ppr[NI], ppc[NI];
#pragma unroll
  for (int n = 0; n < NI; n++) {
    const int it = threadIdx.x + n * BS;
    const bool ok = it < NQ * PT;
    const int cq = ok ? it / PT : 0, pp = ok ? it % PT : 0, pr = pp / kPoolTC, pc = pp % kPoolTC;
    pwin[n] = ((2 * pr) * kPoolCC + 2 * pc) * XQ + cq;
    ppr[n] = pr;
    ppc[n] = pc;
    poff[n] = ok ? (((cq + NQ * half) * pool.OH + pr) * pool.OW + pc) * 16 : -1;
  }
  const unsigned pooled_img_bytes = unsigned(g.M / 4) * unsigned(pool.OH) * unsigned(pool.OW) * 16u;

#ifdef INFERA_STEM_TIMING
  const bool timing = __builtin_amdgcn_readfirstlane(wave) == 0 && pair == 0 && xcd == 0;
  unsigned long long tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  for (; tile < t_end; tile += tstep) {
    STEM_T(8)
    const int64_t next = tile + tstep;
    const int oy0 = oy0_n, ox0 = ox0_n, img = img_n;
    tile_origin(next < t_end ? next : tile, img_n, oy0_n, ox0_n);
    const int iy0_n = oy0_n * g.sh - g.pt, ix0_n = ox0_n * g.sw - g.pl;
    const float *image_n = X + int64_t(img_n) * g.C * g.H * g.W;

    f32x16 acc[PW];
#pragma unroll
    for (int p = 0; p < PW; p++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[p][i] = 0.f;
    const uint2 *pb[PW];
#pragma unroll
    for (int p = 0; p < PW; p++) pb[p] = patch + lbase[p];
    // a step = one k-block for BOTH pixel tiles: eight cut patch words per tile (even half: kx 0 2 4 6, odd half: kx 1 3 5 7) and the k-block's
    // three weight fragments, all fetched one k-block ahead.  The order is PINNED with sched_barriers: fragments of this k-block assembled ->
    // the next k-block's LDS reads issued -> twelve matrix instructions (the two tiles' accumulators alternate, so consecutive ones are
    // independent).  Left to itself the scheduler sinks the reads below the matrix instructions and waits for them at once.
    // The patch words are read with ds_read_b64 (two LDS cycles per wave instruction, 64 banks), written as inline assembly: the compiler fuses
    // the two adjacent words of a lane into ds_read2_b64, which the LDS serves at a quarter of that rate (eight cycles per instruction: 32 banks,
    // 16-lane groups) -- sixteen of them per k-block and wave kept the LDS, not the matrix cores, busy (MI355X_MICROARCH.md, LDS table;
    // profiles/r04_stem_lds_ab.txt).  The counter wait is written by hand as well (wait_words), with the words as operands so that no use moves
    // above it.  What the constraints can and cannot promise (ADVICE r4 / VERDICT r5 item 7): the destinations are EARLY-CLOBBER ("=&v": never the
    // address register of a read that is still to be issued) and every use of a word goes through wait_words' "+v" operand, so nothing can read a
    // word before the wait; what C++-level asm cannot forbid is a register COPY of a word between its read and the wait (the copy would carry
    // the old contents).  hipcc 7.2 (clang 20, the version the Makefile pins and __graft_entry__.build() prints) makes none -- the k loop has no
    // spill and no v_mov of a word register before its wait -- and tests/test_stem_pool_gpu.py (fused stem == INFERA_STEM_SPLIT=0's exact-fp32
    // kernels' plan within the split arithmetic's bound, and == itself across wave counts bit for bit) is the gate a toolchain bump has to pass.
    using u64w = unsigned long long;
    u64w w[PW][8];
    u32x4_t a3[2][3];
    auto fetch_words = [&](int kb, int p) {
      const uint2 *row = pb[p] + (h ? soff[2 * kb + 1] : soff[2 * kb]);
      const unsigned a0 = unsigned(size_t((const __attribute__((address_space(3))) void *)row)), a1 = a0 + unsigned(pg.HALF) * 8u;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(w[p][e]) : "v"(a0), "n"(8 * e));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(w[p][4 + e]) : "v"(a1), "n"(8 * e));
      }
    };
    // (the weight fragments the same way: a read the compiler counts would make it wait for "all but the last three" LDS reads before the matrix
    //  instructions -- that is, for every patch word just issued)
    const unsigned wfrag_a = unsigned(size_t((const __attribute__((address_space(3))) void *)wfrag));
    auto fetch_weights = [&](int kb, int buf) {
#pragma unroll
      for (int k = 0; k < 3; k++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(a3[buf][k]) : "v"(wfrag_a), "n"((kb * 3 + k) * 1024));
    };
    auto wait_words = [&](int buf) {
      if constexpr (PW == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[0][3]), "+v"(w[0][4]), "+v"(w[0][5]), "+v"(w[0][6]), "+v"(w[0][7]),
                       "+v"(w[PW - 1][0]), "+v"(w[PW - 1][1]), "+v"(w[PW - 1][2]), "+v"(w[PW - 1][3]), "+v"(w[PW - 1][4]), "+v"(w[PW - 1][5]),
                       "+v"(w[PW - 1][6]), "+v"(w[PW - 1][7]), "+v"(a3[buf][0]), "+v"(a3[buf][1]), "+v"(a3[buf][2]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[0][3]), "+v"(w[0][4]), "+v"(w[0][5]), "+v"(w[0][6]), "+v"(w[0][7]),
                       "+v"(a3[buf][0]), "+v"(a3[buf][1]), "+v"(a3[buf][2]));
    };
#pragma unroll
    for (int p = 0; p < PW; p++) fetch_words(0, p);
    fetch_weights(0, 0);
    STEM_T(9)
#pragma unroll
    for (int kb = 0; kb < KBC; kb++) {
      const int kc = kb & 1;
      u32x4_t bh[PW], bm[PW], bl[PW];
      wait_words(kc);
#pragma unroll
      for (int p = 0; p < PW; p++)
#pragma unroll
        for (int i = 0; i < 4; i++) {  // elements 2i, 2i + 1 of the fragment: hi halves, mid halves, lo halves of two cut words
          const unsigned x0 = unsigned(w[p][2 * i]), y0 = unsigned(w[p][2 * i] >> 32), x1 = unsigned(w[p][2 * i + 1]), y1 = unsigned(w[p][2 * i + 1] >> 32);
          bh[p][i] = __builtin_amdgcn_perm(x1, x0, 0x05040100u);
          bm[p][i] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
          bl[p][i] = __builtin_amdgcn_perm(y1, y0, 0x05040100u);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 1 < KBC) {
#pragma unroll
        for (int p = 0; p < PW; p++) fetch_words(kb + 1, p);
        fetch_weights(kb + 1, kc ^ 1);
      }
      // the next tile's patch words ride along: sixteen (eight) fetches over the eleven k-blocks
#pragma unroll
      for (int sl = 0; sl < PE; sl++)
        if (sl * KBC / PE == kb) pv[sl] = load_slot(sl, image_n, iy0_n, ix0_n);
      __builtin_amdgcn_sched_barrier(0);
      auto B = [](const u32x4_t &v) { return __builtin_bit_cast(bf16x8_s, v); };
#pragma unroll
      for (int p = 0; p < PW; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B(a3[kc][2]), B(bh[p]), acc[p], 0, 0, 0);
#pragma unroll
      for (int p = 0; p < PW; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B(a3[kc][0]), B(bl[p]), acc[p], 0, 0, 0);
#pragma unroll
      for (int p = 0; p < PW; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B(a3[kc][1]), B(bm[p]), acc[p], 0, 0, 0);
#pragma unroll
      for (int p = 0; p < PW; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B(a3[kc][1]), B(bh[p]), acc[p], 0, 0, 0);
#pragma unroll
      for (int p = 0; p < PW; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B(a3[kc][0]), B(bm[p]), acc[p], 0, 0, 0);
#pragma unroll
      for (int p = 0; p < PW; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B(a3[kc][0]), B(bh[p]), acc[p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    STEM_T(0)
    __syncthreads();  // everybody is out of the k loop: the patch region becomes the exchange tile
    STEM_T(1)
    dispatch_act(act.kind, [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
      for (int p = 0; p < PW; p++) {
        const int oy = oy0 + py[p], ox = ox0 + px[p];
        const bool inside = oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = inside ? apply_act_c<KIND>(acc[p][4 * q + j] + bres[q][j], act.a, act.b) : -INFINITY;
          exch[((wave + NW * p) * 32 + r) * XQ + 2 * q + h] = v;
        }
      }
    });
    STEM_T(2)
    __syncthreads();
    STEM_T(3)
    {  // pooling: every thread its (at most two) pooled (pixel, channel quad) items
      const char *base = reinterpret_cast<const char *>(Y) + int64_t(__builtin_amdgcn_readfirstlane(img)) * pooled_img_bytes;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, int(pooled_img_bytes), 0x00020000);
      const int pr0 = (oy0 + pool.pt) >> 1, pc0 = (ox0 + pool.pl) >> 1;
#pragma unroll
      for (int n = 0; n < NI; n++) {
        const f32x4 *win = exch + pwin[n];
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const f32x4 v = win[(i * kPoolCC + j) * XQ];
#pragma unroll
            for (int e = 0; e < 4; e++) m[e] = fmaxf(m[e], v[e]);
          }
        const int voff = (poff[n] >= 0 && pr0 + ppr[n] < pool.OH && pc0 + ppc[n] < pool.OW) ? poff[n] + (pr0 * pool.OW + pc0) * 16 : -1;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, m), rs, voff, 0, 0);
      }
    }
    STEM_T(4)
    __syncthreads();  // the exchange tile has been read: the region is the patch again
    STEM_T(5)
    park_patch(pv);   // (unconditional: the last tile parks its own re-fetched patch)
    STEM_T(6)
    __syncthreads();
    STEM_T(7)
  }
#ifdef INFERA_STEM_TIMING
  if (timing && lane == 0)
    for (int i = 0; i < 10; i++) g_stem_phase[half][i] = tsum[i];
#endif
}

// Depthwise convolution (groups == C == M) in channel-quad planes: HBM-bound, no matrix cores.  One thread per
// (n, channel quad, oh, ow): every tap is one 16-byte load (consecutive lanes walk a plane row) times one
// 16-byte weight quad [c/4][tap][4] that the whole wave shares.
// (Tried for 3x3: one thread per strip of four outputs with the shared input quads held in registers and branch-free
// clamped loads -- strips along a row: 4.39 ms, strips down a column: 4.20 ms, this kernel: 4.03 ms for the
// MobileNetV2 topology at 256 images.  Fewer load instructions did not pay for a quarter of the threads.)
// Grid: x = plane (n * C/4 + c/4), y = blocks of BS positions inside the plane -- the plane, its channel quad and its
// weights are scalar (per workgroup), the position costs one 32-bit division; the first version decomposed a flat
// 64-bit index per thread (four 64-bit divisions, ~160 VALU instructions per output quad, more than the 36 FMAs).
template <int BS>
__global__ __launch_bounds__(BS) void conv2d_depthwise_cq_kernel(const float *__restrict__ X, const float *__restrict__ Wd,
                                                                 const float *__restrict__ bias, float *__restrict__ Y,
                                                                 ConvGeom g, ActParam act) {
  const int pos = int(blockIdx.y) * BS + int(threadIdx.x), ohw = g.OH * g.OW;
  if (pos >= ohw) return;
  const int64_t plane = blockIdx.x;
  const int c4 = int(plane % (g.C >> 2)), ntaps = g.kh * g.kw;
  const int oh = pos / g.OW, ow = pos - oh * g.OW;
  const f32x4 *x4 = reinterpret_cast<const f32x4 *>(X) + plane * g.H * g.W;
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(Wd) + int64_t(c4) * ntaps;
  f32x4 acc = bias ? reinterpret_cast<const f32x4 *>(bias)[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
  const int iy0 = oh * g.sh - g.pt, ix0 = ow * g.sw - g.pl;
  for (int ky = 0; ky < g.kh; ky++) {
    const int iy = iy0 + ky * g.dh;
    if (iy < 0 || iy >= g.H) continue;
    for (int kx = 0; kx < g.kw; kx++) {
      const int ix = ix0 + kx * g.dw;
      if (ix < 0 || ix >= g.W) continue;
      const f32x4 v = x4[iy * g.W + ix], w = wq[ky * g.kw + kx];
#pragma unroll
      for (int e = 0; e < 4; e++) acc[e] = fmaf(v[e], w[e], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) acc[e] = apply_act(acc[e], act);
  reinterpret_cast<f32x4 *>(Y)[plane * ohw + pos] = acc;
}

// CQ pooling: one thread per (n, channel quad, oh, ow) -- 16 bytes per tap, consecutive lanes walk a plane row.
template <int BS>
__global__ __launch_bounds__(BS) void pool2d_cq_kernel(const float *__restrict__ X, float *__restrict__ Y, int H, int W, int OH, int OW,
                                                       int kh, int kw, int sh, int sw, int pt, int pl, int dh, int dw, bool is_max,
                                                       bool count_pad) {
  // grid: x = plane (n * C/4 + c/4), y = blocks of BS positions (see the depthwise kernel)
  const int pos = int(blockIdx.y) * BS + int(threadIdx.x), ohw = OH * OW;
  if (pos >= ohw) return;
  const int64_t plane = blockIdx.x;
  const int oh = pos / OW, ow = pos - oh * OW;
  const f32x4 *x4 = reinterpret_cast<const f32x4 *>(X) + plane * H * W;
  f32x4 acc = is_max ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < kh; i++) {
    const int iy = oh * sh - pt + i * dh;
    if (iy < 0 || iy >= H) continue;
    for (int j = 0; j < kw; j++) {
      const int ix = ow * sw - pl + j * dw;
      if (ix < 0 || ix >= W) continue;
      const f32x4 v = x4[iy * W + ix];
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
  reinterpret_cast<f32x4 *>(Y)[plane * ohw + pos] = acc;
}

// Max pooling with a compile-time window (3x3/2 after a ResNet stem, 2x2/2 in VGG-style nets, 3x3/1 in inception
// branches): the generic kernel above skips out-of-image taps with a branch per tap, so its loads issue one at a time
// behind each other's s_waitcnt.  For MAX a tap outside the image can simply be clamped onto the nearest row / column --
// that element is inside the same window (pads are smaller than the window, dilation 1), so the maximum is unchanged --
// and the KH x KW sixteen-byte loads of a thread are straight-line code, all in flight together: 1.09 -> 0.92 ms on
// ResNet-18's 64 x 112 x 112 map at 1024 images (4.5 TB/s).  (Two outputs per thread sharing the middle column: 1.19 ms.)
template <int KH, int KW, int BS>
__global__ __launch_bounds__(BS) void pool2d_cq_max_kernel(const float *__restrict__ X, float *__restrict__ Y, int H, int W, int OH,
                                                           int OW, int sh, int sw, int pt, int pl) {
  const int pos = int(blockIdx.y) * BS + int(threadIdx.x);
  if (pos >= OH * OW) return;
  const int64_t plane = blockIdx.x;
  const int oh = pos / OW, ow = pos - oh * OW;
  const f32x4 *x4 = reinterpret_cast<const f32x4 *>(X) + plane * H * W;
  const int iy0 = oh * sh - pt, ix0 = ow * sw - pl;
  f32x4 v[KH][KW];
#pragma unroll
  for (int i = 0; i < KH; i++) {
    const int iy = min(max(iy0 + i, 0), H - 1);
#pragma unroll
    for (int j = 0; j < KW; j++) v[i][j] = x4[iy * W + min(max(ix0 + j, 0), W - 1)];
  }
  f32x4 acc = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int i = 0; i < KH; i++)
#pragma unroll
    for (int j = 0; j < KW; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc[e] = fmaxf(acc[e], v[i][j][e]);
  reinterpret_cast<f32x4 *>(Y)[plane * OH * OW + pos] = acc;
}

__global__ __launch_bounds__(kBlock) void pool2d_kernel(const float *__restrict__ X, float *__restrict__ Y, int64_t total,
                                                       int C, int H, int W, int OH, int OW, int kh, int kw, int sh, int sw,
                                                       int pt, int pl, int dh, int dw, bool is_max, bool count_pad) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t o = int64_t(blockIdx.x) * kBlock + threadIdx.x; o < total; o += stride) {
    const int ow = int(o % OW);
    const int oh = int((o / OW) % OH);
    const int c = int((o / (int64_t(OW) * OH)) % C);
    const int64_t n = o / (int64_t(OW) * OH * C);
    float acc = is_max ? -INFINITY : 0.f;
    int cnt = 0;
    for (int i = 0; i < kh; i++)
      for (int j = 0; j < kw; j++) {
        const int iy = oh * sh - pt + i * dh, ix = ow * sw - pl + j * dw;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float v = X[act_index(false, n, c, iy, ix, C, H, W)];
        acc = is_max ? fmaxf(acc, v) : acc + v;
        cnt++;
      }
    if (!is_max) acc = acc / float(count_pad ? kh * kw : (cnt ? cnt : 1));
    Y[o] = acc;
  }
}

// NCHW: one wave per (n, c) over S contiguous elements.  CQ: one wave per (n, channel quad), 16 bytes per lane per step.
// is_max: GlobalMaxPool (same traversal, max instead of mean).
__global__ __launch_bounds__(kBlock) void global_avgpool_kernel(const float *__restrict__ X, float *__restrict__ Y,
                                                               int64_t nc_total, int C, int S, bool cq, bool is_max) {
  if (cq) {
    // one WAVE per channel-quad plane: lanes stride the plane's S quads (16-byte loads, contiguous across the wave),
    // then a butterfly per component.  (One LANE per plane -- the first version -- walked 200 KB apart from its
    // neighbours: 0.6 TB/s on a 112x112 map, i.e. every squeeze-and-excitation block of an EfficientNet.)
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6, nwaves = (int64_t(gridDim.x) * kBlock) >> 6;
    const int64_t nq = nc_total >> 2;
    for (int64_t o = wave; o < nq; o += nwaves) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(X) + o * S;
      f32x4 acc = is_max ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int i = lane; i < S; i += 64) {
        const f32x4 v = src[i];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = is_max ? fmaxf(acc[j], v[j]) : acc[j] + v[j];
      }
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float other = __shfl_xor(acc[j], sh);
          acc[j] = is_max ? fmaxf(acc[j], other) : acc[j] + other;
        }
      if (lane == 0) reinterpret_cast<f32x4 *>(Y)[o] = is_max ? acc : acc / float(S);
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kBlock) >> 6;
  for (int64_t nc = wave; nc < nc_total; nc += nwaves) {
    const float *src = X + nc * S;
    float acc = is_max ? -INFINITY : 0.f;
    for (int i = lane; i < S; i += 64) acc = is_max ? fmaxf(acc, src[i]) : acc + src[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = is_max ? fmaxf(acc, __shfl_xor(acc, o)) : acc + __shfl_xor(acc, o);
    if (lane == 0) Y[nc] = is_max ? acc : acc / float(S);
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
            ActParam act, bool in_cq, bool out_cq) {
  const int64_t total_pix = rows * g.OH * g.OW;
  if (total_pix <= 0) return;
  const int Mg = g.M / g.groups;
  const size_t lds = size_t(g.C / g.groups) * g.kh * g.kw * sizeof(KEntry);
  const unsigned bx = unsigned((total_pix + 127) / 128);
  if (Mg <= 32) {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 31) / 32)));
    hipLaunchKernelGGL(conv2d_generic_kernel<1>, grid, dim3(kBlock), lds, s, X, Wk, bias, Y, total_pix, g, act, in_cq, out_cq);
  } else if (Mg <= 64) {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 63) / 64)));
    hipLaunchKernelGGL(conv2d_generic_kernel<2>, grid, dim3(kBlock), lds, s, X, Wk, bias, Y, total_pix, g, act, in_cq, out_cq);
  } else {
    dim3 grid(bx, unsigned(g.groups * ((Mg + 127) / 128)));
    hipLaunchKernelGGL(conv2d_generic_kernel<4>, grid, dim3(kBlock), lds, s, X, Wk, bias, Y, total_pix, g, act, in_cq, out_cq);
  }
}

ConvGeom conv2d_patch_geom(const ConvGeom &real) {
  ConvGeom g = real;
  if (real.M % 32 != 0) {
    g.M = (real.M + 31) / 32 * 32;
    g.mvalid = real.M;
  }
  return g;
}

// (pass the padded geometry: conv2d_patch_geom)
bool conv2d_patch_supported(const ConvGeom &g) {
  if (g.groups != 1 || g.M % 32 != 0 || g.M > 128 || g.C > 8 || (g.mvalid > 0 && g.mvalid % 4 != 0)) return false;
  if (int64_t(g.C) * g.H * g.W >= (int64_t(1) << 30)) return false;
  const PatchGeom p = patch_geom(g);
  return p.NE <= kPatchMaxE && p.PR < 32768 && p.PC < 65536 && patch_lds_bytes(g, p) <= 160 * 1024;
}

// ... followed by MaxPool 3x3 / 2 whose windows the 17 x 15 tile covers (pads 0 or 1, no dilation; ceil_mode or not: pooled
// pixels whose windows run past the convolution's output see -inf there, as in the stand-alone kernel)
bool conv2d_patch_pool_supported(const ConvGeom &g, const PoolTail &pool) {
  if (!conv2d_patch_supported(g) || g.mvalid > 0 || g.M > 64 || pool.OH < 1 || pool.OW < 1 || pool.pt < 0 || pool.pt > 1 || pool.pl < 0 || pool.pl > 1)
    return false;
  if ((pool.OH - 1) * 2 - pool.pt >= g.OH || (pool.OW - 1) * 2 - pool.pl >= g.OW) return false;  // a window with no pixel at all
  if (int64_t(g.M) * pool.OH * pool.OW * 4 >= (int64_t(1) << 31)) return false;                  // (one image's pooled planes = one buffer descriptor)
  const PatchGeom p = patch_pool_geom(g, pool);
  return p.NE <= kPatchMaxE && p.PR < 32768 && p.PC < 65536 && patch_pool_lds_bytes(g, p) <= 160 * 1024;
}

size_t conv2d_patch_packed_floats(const ConvGeom &g) {
  const PatchGeom p = patch_geom(g);
  return size_t(p.K8) * (g.M / 32) * 256 + size_t(p.K8) * 8;
}

// [K8][MT][lane][j] = Wt[m = 32mt + (lane&31)][k = 8g + 4*(lane>>5) + j] (zero past C*kh*kw), then the
// per-k patch offsets [K8][h][j] (as int bit patterns), k = (c, ky, kx) in ONNX order
void conv2d_patch_pack(const ConvGeom &g, const float *Wt, float *packed, const PoolTail *pool) {
  const PatchGeom p = pool ? patch_pool_geom(g, *pool) : patch_geom(g);  // (the per-k offsets depend on the patch's row length)
  const int MT = g.M / 32, KK = g.C * g.kh * g.kw;
  for (int grp = 0; grp < p.K8; grp++)
    for (int mt = 0; mt < MT; mt++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 4; j++) {
          const int m = 32 * mt + (lane & 31), k = 8 * grp + 4 * (lane >> 5) + j;
          packed[((size_t(grp) * MT + mt) * 64 + lane) * 4 + j] = k < KK ? Wt[size_t(m) * KK + k] : 0.f;
        }
  int *kt = reinterpret_cast<int *>(packed + size_t(p.K8) * MT * 256);
  for (int k = 0; k < p.K8 * 8; k++) {
    int off = 0;
    if (k < KK) {
      const int c = k / (g.kh * g.kw), rem = k % (g.kh * g.kw), cy = (rem / g.kw) * g.dh, cx = (rem % g.kw) * g.dw;
      off = c * p.PLANE + cy * p.ROWS + (cx % g.sw) * p.HALF + cx / g.sw;
    }
    kt[k] = off;
  }
}

void conv2d_patch_pool(hipStream_t s, const float *X, const float *packed, const float *bias, float *Y, int64_t rows,
                       const ConvGeom &g, ActParam act, const PoolTail &pool, int num_cus) {
  if (rows <= 0) return;
  const PatchGeom p = patch_pool_geom(g, pool);
  if (const int64_t cap = ((int64_t(1) << 31) - 1) / (int64_t(p.tiles_x) * p.tiles_y); rows > cap) {  // (the kernel counts tiles in 32 bits)
    for (int64_t r0 = 0; r0 < rows; r0 += cap)
      conv2d_patch_pool(s, X + r0 * g.C * g.H * g.W, packed, bias, Y + r0 * g.M * pool.OH * pool.OW, std::min(cap, rows - r0), g, act, pool, num_cus);
    return;
  }
  const int64_t ntiles = rows * p.tiles_x * p.tiles_y;
  // 64-feature stems with a compile-time k loop: two half-channel workgroups per CU (INFERA_STEM_POOL2=0: the one-workgroup kernel;
  // INFERA_STEM_POOL2_DESYNC=n: the odd half starts n x 3.5 us late, default 1)
  // (read per launch -- two getenv calls against a millisecond kernel -- so that one process can run both kernels: the bit-identity tests do)
  const int pool2 = getenv("INFERA_STEM_POOL2") ? atoi(getenv("INFERA_STEM_POOL2")) : 1;
  const int desync = getenv("INFERA_STEM_POOL2_DESYNC") ? atoi(getenv("INFERA_STEM_POOL2_DESYNC")) : 1;
  const int cus = num_cus > 0 ? num_cus : 256;
  if (pool2 && g.M == 64 && (p.K8 == 19 || p.K8 == 10) && (g.C * p.PR * p.PC + kPool2Block - 1) / kPool2Block <= kPatchMaxE &&
      2 * patch_pool2_lds_bytes(g, p) <= 160 * 1024 && cus % 8 == 0 && (pool2 >= 2 || ntiles >= int64_t(cus))) {  // (2: always -- tests)
    const size_t lds2 = patch_pool2_lds_bytes(g, p);
    const unsigned grid2 = unsigned(2 * cus);  // a multiple of 16: an even number of workgroups (whole pairs) on every XCD
    auto launch2 = [&](auto kernel) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(kernel, dim3(grid2), dim3(kPool2Block), lds2, s, X, packed, bias, Y, ntiles, g, p, act, pool, desync);
    };
    if (p.K8 == 19) launch2(conv2d_stem_pool2_kernel<19>);
    else launch2(conv2d_stem_pool2_kernel<10>);
    return;
  }
  const size_t lds = patch_pool_lds_bytes(g, p);
  const unsigned grid = unsigned(std::min<int64_t>(ntiles, int64_t(cus)));
  auto launch = [&](auto kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kPoolBlock), lds, s, X, packed, bias, Y, ntiles, g, p, act, pool);
  };
  auto by_k8 = [&](auto mt_tag) {
    constexpr int MT = decltype(mt_tag)::value;
    switch (p.K8) {
      case 19: launch(conv2d_patch_kernel<MT, 19, true>); break;
      case 10: launch(conv2d_patch_kernel<MT, 10, true>); break;
      case 4: launch(conv2d_patch_kernel<MT, 4, true>); break;
      default: launch(conv2d_patch_kernel<MT, 0, true>); break;
    }
  };
  if (g.M == 32) by_k8(std::integral_constant<int, 1>{});
  else by_k8(std::integral_constant<int, 2>{});
}

// ---- bf16 x three parts stem + max-pool (conv2d_stem_split6_kernel) ----
// (Round 4, first try: a different ROW pitch for the 8-byte patch words while the k loop still fetched them with the compiler's ds_read2_b64 --
// no change, profiles/r04_stem_pitch_ab.txt.  What the LDS share was really made of -- the instruction, then the banks -- is in
// stem_split6_geom's and the k loop's comments: profiles/r04_stem_lds_ab.txt, 2186 -> 1895 us.)
bool conv2d_stem_split6_supported(const ConvGeom &g, const PoolTail &pool) {
  if (!conv2d_patch_pool_supported(g, pool) || g.M != 64) return false;
  const PatchGeom p = stem_split6_geom(g, pool);
  // (the k loop reads four consecutive words of each half of a de-interleaved patch row: 7 columns, stride 2)
  return g.kw == 7 && g.sw == 2 && g.dw == 1 && g.sh == 2 && g.dh == 1 && (g.C * g.kh + 1) / 2 == kStemKB && p.HALF >= kPoolCC + 3 &&
         (g.C * p.PR * p.PC + kPool2Block - 1) / kPool2Block <= kPatchMaxE && 2 * stem_split6_lds_bytes(g, p) <= 160 * 1024;
}

size_t conv2d_stem_split6_packed_floats() { return size_t(kStemKB) * 1536; }

void conv2d_stem_split6_pack(const ConvGeom &g, const float *Wt, float *packed) {
  const int KK = g.C * g.kh * g.kw;
  uint16_t *out = reinterpret_cast<uint16_t *>(packed);
  for (int kb = 0; kb < kStemKB; kb++)
    for (int half = 0; half < 2; half++)
      for (int lane = 0; lane < 64; lane++)
        for (int e = 0; e < 8; e++) {
          const int m = 32 * half + (lane & 31), rr = 2 * kb + (lane >> 5), kx = e < 4 ? 2 * e : 2 * (e - 4) + 1;
          const float v = rr < g.C * g.kh && kx < g.kw ? Wt[size_t(m) * KK + size_t(rr) * g.kw + kx] : 0.f;
          uint32_t x, y, z;  // exact truncation cut: v = hi + mid + lo
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
          const size_t base = (size_t(kb) * 2 + half) * 3;  // fragments of 64 lanes x 8 bf16
          out[(base + 0) * 512 + size_t(lane) * 8 + e] = uint16_t(x >> 16);
          out[(base + 1) * 512 + size_t(lane) * 8 + e] = uint16_t(y >> 16);
          out[(base + 2) * 512 + size_t(lane) * 8 + e] = uint16_t(z >> 16);
        }
}

void conv2d_stem_split6(hipStream_t s, const float *X, const float *packed, const float *bias, float *Y, int64_t rows, const ConvGeom &g,
                        ActParam act, const PoolTail &pool, int num_cus) {
  if (rows <= 0) return;
  const PatchGeom p = stem_split6_geom(g, pool);
  if (const int64_t cap = ((int64_t(1) << 31) - 1) / (int64_t(p.tiles_x) * p.tiles_y); rows > cap) {
    for (int64_t r0 = 0; r0 < rows; r0 += cap)
      conv2d_stem_split6(s, X + r0 * g.C * g.H * g.W, packed, bias, Y + r0 * g.M * pool.OH * pool.OW, std::min(cap, rows - r0), g, act, pool, num_cus);
    return;
  }
  const int64_t ntiles = rows * p.tiles_x * p.tiles_y;
  const int desync = getenv("INFERA_STEM_POOL2_DESYNC") ? atoi(getenv("INFERA_STEM_POOL2_DESYNC")) : 1;
  const int cus = std::max(8, (num_cus > 0 ? num_cus : 256) / 8 * 8);  // whole workgroup pairs on each of 8 XCD queues; runs for every batch size
  // INFERA_STEM_WAVES=4|8 (measurement knob): workgroups of four waves with two pixel tiles each, or of eight with one
  static const int waves = getenv("INFERA_STEM_WAVES") && atoi(getenv("INFERA_STEM_WAVES")) == 8 ? 8 : 4;
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv2d_stem_split6_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv2d_stem_split6_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
  }
  if (waves == 8)
    hipLaunchKernelGGL(conv2d_stem_split6_kernel<8>, dim3(unsigned(2 * cus)), dim3(512), stem_split6_lds_bytes(g, p), s, X, packed, bias, Y, ntiles, g, p, act,
                       pool, desync);
  else
    hipLaunchKernelGGL(conv2d_stem_split6_kernel<4>, dim3(unsigned(2 * cus)), dim3(256), stem_split6_lds_bytes(g, p), s, X, packed, bias, Y, ntiles, g, p, act,
                       pool, desync);
}

void conv2d_patch(hipStream_t s, const float *X, const float *packed, const float *bias, float *Y, int64_t rows,
                  const ConvGeom &g, ActParam act, int num_cus) {
  if (rows <= 0) return;
  const PatchGeom p = patch_geom(g);
  if (const int64_t cap = ((int64_t(1) << 31) - 1) / (int64_t(p.tiles_x) * p.tiles_y); rows > cap) {  // (the kernel counts tiles in 32 bits)
    const int64_t out_row = int64_t(g.mvalid > 0 ? g.mvalid : g.M) * g.OH * g.OW;
    for (int64_t r0 = 0; r0 < rows; r0 += cap)
      conv2d_patch(s, X + r0 * g.C * g.H * g.W, packed, bias, Y + r0 * out_row, std::min(cap, rows - r0), g, act, num_cus);
    return;
  }
  const int64_t ntiles = rows * p.tiles_x * p.tiles_y;
  const size_t lds = patch_lds_bytes(g, p);
  const int per_cu = int(std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds)));
  const unsigned grid = unsigned(std::min<int64_t>(ntiles, int64_t(num_cus > 0 ? num_cus : 256) * per_cu));
  auto launch = [&](auto kernel) {
    if (lds > 64 * 1024)  // dynamic LDS beyond 64 KB is opt-in (per device, so not cached here)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, s, X, packed, bias, Y, ntiles, g, p, act, PoolTail{});
  };
  auto by_k8 = [&](auto mt_tag) {
    constexpr int MT = decltype(mt_tag)::value;
    switch (p.K8) {  // the usual 3-channel stems: 7x7 (147 -> 19 groups), 5x5 (75 -> 10), 3x3 (27 -> 4)
      case 19: launch(conv2d_patch_kernel<MT, 19>); break;
      case 10: launch(conv2d_patch_kernel<MT, 10>); break;
      case 4: launch(conv2d_patch_kernel<MT, 4>); break;
      default: launch(conv2d_patch_kernel<MT, 0>); break;
    }
  };
  switch (g.M / 32) {
    case 1: by_k8(std::integral_constant<int, 1>{}); break;
    case 2: by_k8(std::integral_constant<int, 2>{}); break;
    case 3: by_k8(std::integral_constant<int, 3>{}); break;
    default: by_k8(std::integral_constant<int, 4>{}); break;
  }
}

bool conv2d_depthwise_supported(const ConvGeom &g) { return g.groups == g.C && g.M == g.C && g.C % 4 == 0; }

// [C/4][tap][4] <- Wt[m][0][ky][kx]
void conv2d_depthwise_pack(const ConvGeom &g, const float *Wt, float *packed) {
  const int ntaps = g.kh * g.kw;
  for (int c = 0; c < g.C; c++)
    for (int t = 0; t < ntaps; t++) packed[(size_t(c >> 2) * ntaps + t) * 4 + (c & 3)] = Wt[size_t(c) * ntaps + t];
}

void conv2d_depthwise(hipStream_t s, const float *X, const float *packed, const float *bias, float *Y, int64_t rows,
                      const ConvGeom &g, ActParam act) {
  const int64_t planes = rows * (g.C / 4);
  const int ohw = g.OH * g.OW;
  if (planes <= 0 || ohw <= 0) return;
  if (ohw <= 64) hipLaunchKernelGGL(conv2d_depthwise_cq_kernel<64>, dim3(unsigned(planes), unsigned((ohw + 63) / 64)), dim3(64), 0, s, X, packed, bias, Y, g, act);
  else hipLaunchKernelGGL(conv2d_depthwise_cq_kernel<256>, dim3(unsigned(planes), unsigned((ohw + 255) / 256)), dim3(256), 0, s, X, packed, bias, Y, g, act);
}

// C and M multiples of 32, or (channel-quad tensors) of 4: those run with zero-padded weights, see conv2d_tiled_geom
bool conv2d_tiled_supported(const ConvGeom &g) {
  return g.groups == 1 && g.C % 4 == 0 && g.M % 4 == 0 && g.kh * g.kw <= 64 /* per-lane tap mask */ &&
         int64_t(g.H) * g.W * g.C < (int64_t(1) << 30);
}

ConvGeom conv2d_tiled_geom(const ConvGeom &real) {
  ConvGeom g = real;
  if (real.mvalid > 0 || (real.C % 32 == 0 && real.M % 32 == 0)) return g;  // dense layers arrive padded already
  g.C = (real.C + 31) / 32 * 32;
  g.M = (real.M + 31) / 32 * 32;
  g.kvalid = real.C;
  g.mvalid = real.M;
  g.padc = 1;
  return g;
}

size_t conv2d_tiled_packed_floats(const ConvGeom &g) { return size_t(g.kh) * g.kw * g.C * g.M; }

void conv2d_tiled_pack(const ConvGeom &g, const float *Wt, float *packed) {
  // 32-channel chunks in the kernel's stage order: channel block (S chunks, S = 2 when C % 64 == 0) outermost,
  // then the filter tap, then the chunk inside the block
  const int CC = g.C / 32, MTtot = g.M / 32, ntaps = g.kh * g.kw, S = g.C % 64 == 0 ? 2 : 1;
  for (int tap = 0; tap < ntaps; tap++)
    for (int cc = 0; cc < CC; cc++)
      for (int mt = 0; mt < MTtot; mt++)
        for (int q = 0; q < 4; q++)
          for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 4; j++) {
              const int m = 32 * mt + (lane & 31), c = 32 * cc + 8 * q + 4 * (lane >> 5) + j;
              const size_t chunk = (size_t(cc / S) * ntaps + tap) * S + cc % S;
              packed[((chunk * MTtot + mt) * 4 + q) * 256 + size_t(lane) * 4 + j] = Wt[(size_t(m) * g.C + c) * ntaps + tap];
            }
}

#ifdef INFERA_CONV_PROBES
namespace {
void dump_stamps() {
  unsigned long long st[2][12];
  if (hipDeviceSynchronize() != hipSuccess) return;
  if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_conv_stamps), sizeof st) != hipSuccess) return;
  for (int k = 0; k < 2; k++)
    if (st[k][7]) {
      const double w = double(st[k][7]);
      fprintf(stderr, "[conv stamps MT=%d] waves=%.0f avg cycles: index %.0f first-loads %.0f lds+barrier %.0f loop %.0f epi-loads %.0f epi-issue %.0f drain %.0f | in loop: vmcnt-wait %.0f lds-store %.0f barrier %.0f\n",
              k ? 4 : 2, w, st[k][0] / w, st[k][1] / w, st[k][2] / w, st[k][3] / w, st[k][4] / w, st[k][5] / w, st[k][6] / w,
              st[k][8] / w, st[k][9] / w, st[k][10] / w);
    }
  unsigned long long zero[2][12] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), zero, sizeof zero);
}
}  // namespace
#endif

void conv2d_tiled(hipStream_t s, const float *X, const float *packed, const float *bias, const float *residual, float *Y,
                  int64_t rows, const ConvGeom &g, ActParam act) {
  const int64_t total_pix = rows * g.OH * g.OW;
  if (total_pix <= 0) return;
  if (total_pix >= (int64_t(1) << 31)) {  // the kernels decompose pixel indices in 32 bits
    const int64_t cap = ((int64_t(1) << 31) - 1) / (int64_t(g.OH) * g.OW);
    const int64_t in_row = g.kvalid > 0 && g.H == 1 && g.W == 1 && !g.padc ? g.kvalid : int64_t(g.padc ? g.kvalid : g.C) * g.H * g.W;
    const int64_t out_row = int64_t(g.mvalid > 0 ? g.mvalid : g.M) * g.OH * g.OW;
    for (int64_t r0 = 0; r0 < rows; r0 += cap)
      conv2d_tiled(s, X + r0 * in_row, packed, bias, residual ? residual + r0 * out_row : nullptr, Y + r0 * out_row, std::min(cap, rows - r0), g, act);
    return;
  }
  const unsigned bx = unsigned((total_pix + 127) / 128);
  auto launch = [&](auto kernel, int mt) {
    hipLaunchKernelGGL(kernel, dim3(bx, unsigned(g.M / (32 * mt))), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, g, act, 0u);
  };
  // feature tiles per workgroup: the largest of 4, 3, 2, 1 that divides M / 32 (ResNet: 2 or 4; MobileNet-style
  // widths such as 96, 160, 576, 960 take 3, 1, 3, 3)
  const int m32 = g.M / 32;
  int mt_pick = m32 % 4 == 0 ? 4 : m32 % 3 == 0 ? 3 : m32 % 2 == 0 ? 2 : 1;
  // Few, long workgroups leave a tail: ResNet's 512-channel 3x3 layers are 1568 workgroups of 128 features x 72 stages
  // on 512 slots -- three full rounds and a fourth that is 6 % full yet lasts a whole lone-workgroup time (13 % of the
  // launch).  Below ~8 rounds 64-feature tiles (twice the workgroups, a quarter of the tail) win on stride-1 3x3 layers:
  // 1.91 -> 1.79 ms (512 channels), 1.80 -> 1.76 (256); stride-2 and 1x1 layers lose (more slices re-gather more) and stay.
  if (mt_pick == 4 && g.sh == 1 && g.sw == 1 && g.kh * g.kw > 1 && !g.padc && g.mvalid == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int cus[64] = {};
    if (!cus[dev & 63]) (void)hipDeviceGetAttribute(&cus[dev & 63], hipDeviceAttributeMultiprocessorCount, dev);
    if (int64_t(bx) * (m32 / 4) < int64_t(8) * 2 * std::max(1, cus[dev & 63])) mt_pick = 2;
  }
  const bool wide = mt_pick == 4, deep = g.C % 64 == 0;
#ifdef INFERA_CONV_PROBES
  static const int probe = getenv("INFERA_CONV_PROBE") ? atoi(getenv("INFERA_CONV_PROBE")) : 0;
  static int launches = 0;
  if (probe == 5 && ++launches % 38 == 0) dump_stamps();
  if (probe && deep) {
    switch (probe * 2 + (wide ? 1 : 0)) {
      case 2: return launch(conv2d_tiled_kernel<2, 2, 1>, 2);
      case 3: return launch(conv2d_tiled_kernel<4, 2, 1>, 4);
      case 4: return launch(conv2d_tiled_kernel<2, 2, 2>, 2);
      case 5: return launch(conv2d_tiled_kernel<4, 2, 2>, 4);
      case 6: return launch(conv2d_tiled_kernel<2, 2, 3>, 2);
      case 7: return launch(conv2d_tiled_kernel<4, 2, 3>, 4);
      case 8: return launch(conv2d_tiled_kernel<2, 2, 4>, 2);
      case 9: return launch(conv2d_tiled_kernel<4, 2, 4>, 4);
      case 12: return launch(conv2d_tiled_kernel<2, 2, 6>, 2);
      case 13: return launch(conv2d_tiled_kernel<4, 2, 6>, 4);
      case 10: return launch(conv2d_tiled_kernel<2, 2, 5>, 2);
      case 11: return launch(conv2d_tiled_kernel<4, 2, 5>, 4);
    }
  }
#endif
  if (g.padc) {  // channel counts padded to 32 (g = conv2d_tiled_geom(real))
    if (wide) deep ? launch(conv2d_tiled_kernel<4, 2, 0, 2>, 4) : launch(conv2d_tiled_kernel<4, 1, 0, 2>, 4);
    else if (mt_pick == 3) deep ? launch(conv2d_tiled_kernel<3, 2, 0, 2>, 3) : launch(conv2d_tiled_kernel<3, 1, 0, 2>, 3);
    else if (mt_pick == 2) deep ? launch(conv2d_tiled_kernel<2, 2, 0, 2>, 2) : launch(conv2d_tiled_kernel<2, 1, 0, 2>, 2);
    else deep ? launch(conv2d_tiled_kernel<1, 2, 0, 2>, 1) : launch(conv2d_tiled_kernel<1, 1, 0, 2>, 1);
    return;
  }
  if (g.mvalid > 0) {  // dense layer
    if (wide) deep ? launch(conv2d_tiled_kernel<4, 2, 0, 1>, 4) : launch(conv2d_tiled_kernel<4, 1, 0, 1>, 4);
    else if (mt_pick == 3) deep ? launch(conv2d_tiled_kernel<3, 2, 0, 1>, 3) : launch(conv2d_tiled_kernel<3, 1, 0, 1>, 3);
    else if (mt_pick == 2) deep ? launch(conv2d_tiled_kernel<2, 2, 0, 1>, 2) : launch(conv2d_tiled_kernel<2, 1, 0, 1>, 2);
    else deep ? launch(conv2d_tiled_kernel<1, 2, 0, 1>, 1) : launch(conv2d_tiled_kernel<1, 1, 0, 1>, 1);
    return;
  }
  // Weight-stationary persistent kernel when one M-slice of the packed weights fits in LDS (64-channel 3x3 layers, 1x1
  // downsamples): no per-stage weight slab, no barrier in the main loop, the stage stream runs through tile boundaries.
  // INFERA_CONV_WS: 0 = never, 1 (default) = when the launch has enough tiles to fill the persistent grid, 2 = whenever
  // the weights fit (tests).  Read per launch so one process can compare the two kernels.
  const char *ws_env = getenv("INFERA_CONV_WS");
  const int ws_mode = ws_env ? atoi(ws_env) : 1;
  if (ws_mode == 2 || ((ws_mode == 1 || ws_mode == 3) && total_pix >= 32 * 2048)) {
    constexpr size_t kWsLdsBytes = 160 * 1024 - 256;
    const size_t slice32 = size_t(g.kh) * g.kw * g.C * 32 * sizeof(float);  // packed weights of 32 output features
    auto launch_ws = [&](auto kernel, int mt, int nw) {
      const size_t lds = slice32 * mt;
      static std::atomic<uint64_t> attr_done{0};
      int dev = 0;
      (void)hipGetDevice(&dev);
      static int cus[64] = {};
      if (!cus[dev & 63]) (void)hipDeviceGetAttribute(&cus[dev & 63], hipDeviceAttributeMultiprocessorCount, dev);
      if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {  // once per kernel instantiation and device
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(kWsLdsBytes));
        attr_done.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
      }
      const unsigned slices = unsigned(g.M / (32 * mt));
      const int64_t ntiles = (total_pix + 31) / 32;
      unsigned gx = unsigned(std::max(1, cus[dev & 63] / int(slices)));
      gx = unsigned(std::min<int64_t>(gx, (ntiles + nw - 1) / nw));
      hipLaunchKernelGGL(kernel, dim3(gx, slices), dim3(unsigned(nw) * 64), lds, s, X, packed, bias, residual, Y, total_pix, g, act);
    };
    if (m32 % 4 == 0 && slice32 * 4 <= kWsLdsBytes) return deep ? launch_ws(conv2d_ws_kernel<4, 2, 8>, 4, 8) : launch_ws(conv2d_ws_kernel<4, 1, 8>, 4, 8);
    if (m32 % 2 == 0 && slice32 * 2 <= kWsLdsBytes) return deep ? launch_ws(conv2d_ws_kernel<2, 2, 8>, 2, 8) : launch_ws(conv2d_ws_kernel<2, 1, 8>, 2, 8);
    // 128-channel 3x3 layers: 32-feature slices still fit (144 KB); twice the gathers per MFMA of the 64-feature form, yet
    // 1.76-1.78 ms against the tiled kernel's 1.79-1.81 per 236.8 GFLOP layer, and 0.94 against 0.98 ms on the stride-2 entry
    if (deep && slice32 <= kWsLdsBytes && ws_mode != 3) return launch_ws(conv2d_ws_kernel<1, 2, 8>, 1, 8);
  }
  // Tail split (round 3): 128-feature workgroups sit two per CU (64 KB of LDS each); a launch of a little more than a whole number of
  // rounds -- ResNet's 256 -> 512 stride-2 entry: 1568 workgroups on 512 slots -- ends in a round that is 6 % full yet lasts a whole
  // lone-workgroup time.  The pixel blocks of that last partial round run as a second launch of 32-feature tiles instead (four times the
  // workgroups, a quarter of the work each; same sums per output element -> bit-identical).  64-feature tiles for the WHOLE layer lose
  // on stride-2 layers (more feature slices re-gather more: 1.03 -> 1.18 ms); this re-gathers only the tail's few pixel blocks.
  if (wide && deep && g.kh * g.kw > 1) {  // (1x1 layers: measured neutral -- 168 us whole, 155 + 13 split)
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int cus2[64] = {};
    if (!cus2[dev & 63]) (void)hipDeviceGetAttribute(&cus2[dev & 63], hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t slots = int64_t(2) * std::max(1, cus2[dev & 63]), slices = m32 / 4, wgs = int64_t(bx) * slices;
    const int64_t rounds = wgs / slots, rem = wgs - rounds * slots;
    const char *tse = getenv("INFERA_CONV_TAIL_SPLIT");  // (read per launch: one process compares both forms in the tests)
    const bool split_on = !(tse && atoi(tse) == 0);
    if (split_on && rounds >= 1 && rounds <= 6 && rem > 0 && rem * 4 <= slots) {  // a short last round
      const unsigned bx_main = unsigned(rounds * slots / slices), bx_tail = bx - bx_main;
      if (bx_main > 0 && bx_tail > 0) {
        hipLaunchKernelGGL((conv2d_tiled_kernel<4, 2>), dim3(bx_main, unsigned(slices)), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, g, act, 0u);
        hipLaunchKernelGGL((conv2d_tiled_kernel<1, 2>), dim3(bx_tail, unsigned(m32)), dim3(kBlock), 0, s, X, packed, bias, residual, Y, total_pix, g, act, bx_main);
        return;
      }
    }
  }
  if (wide && deep) launch(conv2d_tiled_kernel<4, 2>, 4);
  else if (wide) launch(conv2d_tiled_kernel<4, 1>, 4);
  else if (mt_pick == 3) deep ? launch(conv2d_tiled_kernel<3, 2>, 3) : launch(conv2d_tiled_kernel<3, 1>, 3);
  else if (mt_pick == 2) deep ? launch(conv2d_tiled_kernel<2, 2>, 2) : launch(conv2d_tiled_kernel<2, 1>, 2);
  else deep ? launch(conv2d_tiled_kernel<1, 2>, 1) : launch(conv2d_tiled_kernel<1, 1>, 1);
}

void pool2d(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int H, int W, int OH, int OW, int kh, int kw,
            int sh, int sw, int pt, int pl, int dh, int dw, bool is_max, bool count_pad, bool cq) {
  const int64_t total = rows * C * OH * OW;
  if (total <= 0) return;
  if (cq) {  // the plan guarantees C % 4 == 0 in CQ mode
    const int64_t planes = rows * (C / 4);
    const int ohw = OH * OW;
    // compile-time windows for max pooling (clamped, branch-free loads); INFERA_POOL_FAST=0 keeps the generic kernel
    static const bool fast = [] { const char *e = getenv("INFERA_POOL_FAST"); return !e || atoi(e) != 0; }();
    if (fast && is_max && dh == 1 && dw == 1 && pt < kh && pl < kw && (OH - 1) * sh - pt < H && (OW - 1) * sw - pl < W) {
      auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(unsigned(planes), unsigned((ohw + 255) / 256)), dim3(256), 0, s, X, Y, H, W, OH, OW, sh, sw, pt, pl);
      };
      if (kh == 3 && kw == 3) { go(pool2d_cq_max_kernel<3, 3, 256>); return; }
      if (kh == 2 && kw == 2) { go(pool2d_cq_max_kernel<2, 2, 256>); return; }
    }
    if (ohw <= 64) hipLaunchKernelGGL(pool2d_cq_kernel<64>, dim3(unsigned(planes), unsigned((ohw + 63) / 64)), dim3(64), 0, s, X, Y, H, W, OH, OW, kh, kw, sh, sw, pt, pl, dh, dw, is_max, count_pad);
    else hipLaunchKernelGGL(pool2d_cq_kernel<256>, dim3(unsigned(planes), unsigned((ohw + 255) / 256)), dim3(256), 0, s, X, Y, H, W, OH, OW, kh, kw, sh, sw, pt, pl, dh, dw, is_max, count_pad);
    return;
  }
  hipLaunchKernelGGL(pool2d_kernel, dim3(grid_for(total)), dim3(kBlock), 0, s, X, Y, total, C, H, W, OH, OW, kh, kw, sh, sw, pt,
                     pl, dh, dw, is_max, count_pad);
}

void global_avgpool(hipStream_t s, const float *X, float *Y, int64_t rows, int C, int S, bool cq, bool is_max) {
  const int64_t nc = rows * C;
  if (nc <= 0) return;
  hipLaunchKernelGGL(global_avgpool_kernel, dim3(grid_for(cq ? nc / 4 * 64 : nc * 64)), dim3(kBlock), 0, s, X, Y, nc, C, S, cq, is_max);
}

}  // namespace infera_hip::kern
