// mlp_fused.hip -- whole-chain fused MLP  Y = act3(act2(act1(X.W1+b1).W2+b2).W3+b3)  for gfx950.
//
// This is the kernel behind BASELINE config C2/C3 (128 -> 256 -> 64 -> 1 over a 10M-row table):
// 98,432 flop and 516 B of HBM traffic per table row, i.e. bound by the exact-fp32 matrix cores
// (157.3 TFLOP/s -> 1.60e9 rows/s), not by HBM (8 TB/s -> 15.5e9 rows/s).  The design goal is
// therefore: keep v_mfma_f32_32x32x2_f32 issuing back to back, and touch HBM exactly once per
// input byte.
//
// Structure (one persistent workgroup per CU, 4 waves = one per SIMD, no barrier in the main loop):
//   * A wave owns 32 table rows at a time.  Everything is computed TRANSPOSED,
//         H^T[feature, row] = W^T[feature, k] * H_prev^T[k, row],
//     so table rows sit on the MFMA N axis (lane&31) and features on M.  With that choice the
//     32x32 accumulator layout of layer L (lane -> row, register -> feature, lane half -> +4)
//     is ALREADY a valid B-operand layout for layer L+1: accumulator register i of lane half h
//     holds feature 8*(i>>2)+4h+(i&3), and one MFMA k-step consumes the pair {h=0, h=1}.
//     Activations never leave the register file, never touch LDS, never get shuffled.
//   * The price is a fixed permutation of the fp32 summation order inside every group of 8
//     k-indices (0,4,1,5,2,6,3,7).  Weights are pre-packed on the host ("fragment-major") so that
//     lane l finds its A operands for 4 consecutive k-steps in one 16-byte word:
//         packed[((g*MT + mt)*64 + lane)*4 + j] = W[8g + 4*(lane>>5) + j][32*mt + (lane&31)].
//   * LDS (160 KiB) holds all layer-1 fragments (128 KiB for C2), as many layer-2 fragment groups
//     as still fit (NL2 of G1; the rest streams from L2 -- every CU reads the same bytes, so they
//     stay L2/MALL resident), the bias quads and the layer-3 weights.
//   * The unrolled tile body is a stream of "units": one 16-byte fragment load + the 4 MFMAs that
//     consume it, with a register ring fetching P units ahead and a sched_barrier per unit (without
//     it hipcc hoists hundreds of loads to the top of the unrolled body and spills 785 VGPRs).
//   * bias + activation of layer L are applied lazily to each accumulator quad right before it is
//     consumed as a B operand of layer L+1 (VALU work in the MFMA shadow).
//   * A narrow last layer (D3 <= 4, e.g. the C2 regression head) does not go to the matrix cores --
//     a 32-wide MFMA tile would be 31/32 padding, 4 % of all MFMA cycles -- but is evaluated on the
//     VALU from the layer-2 accumulators of the PREVIOUS tile, sliced into the first units of the
//     next tile's layer 1 so that it hides in the MFMA shadow.
//   * X is read with one 16-byte load per lane per 8 k-indices directly in B-fragment shape; the
//     next tile's rows are requested right after this tile's last layer-2 L2 load, so nothing in
//     this tile ever queues behind HBM latency on the in-order vmcnt counter.
#include <atomic>
#include <cstdlib>

#include "device_common.hpp"

namespace infera_hip::kern {

namespace {

constexpr int kLdsBytes = 160 * 1024;

// P1/P2: fragment prefetch depth (units) for layer 1 / layer 2.  NL2: layer-2 fragment groups kept
// in LDS (-1 = as many as fit).  L3V: evaluate the last layer on the VALU (requires D3 <= 4).
// NR2: layer-2 fragment units pinned in REGISTERS for the lifetime of the wave (-1 = all that do not fit
// in LDS): at one wave per SIMD the kernel owns 512 registers per lane and the tile body needs ~360.
template <int D0_, int D1_, int D2_, int D3_, int A1_, int A2_, int A3_, int WAVES_, int P1_ = 3, int P2_ = 16, int NL2_ = -1,
          int NR2_ = 0, bool L3V_ = (D3_ <= 4)>
struct Cfg {
  static constexpr int D0 = D0_, D1 = D1_, D2 = D2_, D3 = D3_;
  static constexpr int A1 = A1_, A2 = A2_, A3 = A3_, WAVES = WAVES_, P1 = P1_, P2 = P2_;
  static constexpr bool L3V = L3V_;
  static_assert(D0 % 8 == 0 && D1 % 32 == 0 && D2 % 32 == 0 && D3 >= 1 && D3 <= 32, "unsupported chain shape");
  static_assert(!L3V || D3 <= 4, "VALU last layer is for narrow heads only");
  static constexpr int G0 = D0 / 8, G1 = D1 / 8, G2 = D2 / 8;  // groups of 4 k-steps per layer input
  static constexpr int MT1 = D1 / 32, MT2 = D2 / 32, MT3 = 1;  // 32-wide output tiles per layer
  static constexpr int GRP2 = MT2 * 256;                       // floats per layer-2 fragment group
  // ---- packed blob (global memory), floats ----
  static constexpr int OFF_W1 = 0, N_W1 = G0 * MT1 * 256;
  static constexpr int OFF_W2 = OFF_W1 + N_W1, N_W2 = G1 * GRP2;
  static constexpr int OFF_SMALL = OFF_W2 + N_W2;
  //   small block: bias quads [mt][rg][h][4] for layers 1,2; then layer 3 (fragments, or VALU weight
  //   quads [kt][rg][h][m][4]); then layer-3 bias (quads, or D3 plain floats padded to 4)
  static constexpr int S_B1 = 0, N_B1 = MT1 * 32;
  static constexpr int S_B2 = S_B1 + N_B1, N_B2 = MT2 * 32;
  static constexpr int S_W3 = S_B2 + N_B2, N_W3 = L3V ? MT2 * 32 * D3 : G2 * MT3 * 256;
  static constexpr int S_B3 = S_W3 + N_W3, N_B3 = L3V ? 4 : MT3 * 32;
  static constexpr int N_SMALL = S_B3 + N_B3;
  static constexpr int N_TOTAL = OFF_SMALL + N_SMALL;
  // ---- LDS image: [W1][first NL2 groups of W2][small] ----
  static constexpr int FIT2 = (kLdsBytes / 4 - N_W1 - N_SMALL) / GRP2;
  static_assert(FIT2 >= 0, "layer-1 fragments + small block exceed 160 KiB of LDS");
  static constexpr int NL2 = NL2_ < 0 ? (FIT2 > G1 ? G1 : FIT2) : NL2_;
  static_assert(NL2 <= FIT2 && NL2 <= G1, "requested LDS share of layer 2 does not fit");
  static constexpr int NR2 = NR2_ < 0 ? (G1 - NL2) * MT2 : NR2_;  // units held in registers
  static_assert(NR2 <= (G1 - NL2) * MT2, "more register-resident units than layer 2 has outside LDS");
  static constexpr int L_W1 = 0, L_W2 = N_W1, L_SMALL = N_W1 + NL2 * GRP2;
  static constexpr int N_LDS = L_SMALL + N_SMALL;
  static_assert(N_LDS * 4 <= kLdsBytes && (L_SMALL % 4) == 0, "LDS image exceeds 160 KiB");
};

template <class C>
__global__ __launch_bounds__(C::WAVES * 64) void mlp3_kernel(const float *__restrict__ X, const float *__restrict__ packed,
                                                            float *__restrict__ Y, int64_t rows) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- stage the LDS image once per workgroup (coalesced 16 B loads) ----
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(packed);
    f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
    for (int i = threadIdx.x; i < C::L_SMALL / 4; i += C::WAVES * 64) dst[i] = src[i];  // W1 + first NL2 groups of W2
    const f32x4 *ssrc = reinterpret_cast<const f32x4 *>(packed + C::OFF_SMALL);
    f32x4 *sdst = reinterpret_cast<f32x4 *>(lds + C::L_SMALL);
    for (int i = threadIdx.x; i < C::N_SMALL / 4; i += C::WAVES * 64) sdst[i] = ssrc[i];
  }
  __syncthreads();

  const f32x4 *w1 = reinterpret_cast<const f32x4 *>(lds + C::L_W1) + lane;
  const f32x4 *w2l = reinterpret_cast<const f32x4 *>(lds + C::L_W2) + lane;
  const float *small = lds + C::L_SMALL;
  const f32x4 *b1 = reinterpret_cast<const f32x4 *>(small + C::S_B1) + h;
  const f32x4 *b2 = reinterpret_cast<const f32x4 *>(small + C::S_B2) + h;
  const f32x4 *w3 = reinterpret_cast<const f32x4 *>(small + C::S_W3) + (C::L3V ? h * C::D3 : lane);
  const float *b3 = small + C::S_B3;
  const f32x4 *w2g_base = reinterpret_cast<const f32x4 *>(packed + C::OFF_W2) + lane;

  const int64_t ntiles = (rows + 31) >> 5;
  const int64_t tstride = int64_t(gridDim.x) * C::WAVES;
  int64_t tile = int64_t(blockIdx.x) * C::WAVES + wave;
  if (tile >= ntiles) return;

  auto load_x = [&](f32x4(&x)[C::G0], int64_t t) {
    int64_t row = (t << 5) + r;
    if (row >= rows) row = rows - 1;  // tail rows recompute the last row; their stores are masked
    const f32x4 *p = reinterpret_cast<const f32x4 *>(X + row * C::D0 + 4 * h);
#pragma unroll
    for (int g = 0; g < C::G0; g++) x[g] = p[2 * g];
  };

  constexpr int U1 = C::G0 * C::MT1, P1 = C::P1;  // LDS fragments: ~128-cycle latency, unit = 256 cycles
  constexpr int U2 = C::G1 * C::MT2, P2 = C::P2;  // layer-2 ring depth (L2 fragments need ~1-2k cycles)
  constexpr int U2L = C::NL2 * C::MT2;            // units [0, U2L) come from LDS,
  constexpr int U2R = U2L + C::NR2;               //       [U2L, U2R) from registers, [U2R, U2) from L2
  constexpr int U3 = C::G2, P3 = 2;
  constexpr int NQ = C::MT2 * 4;                  // accumulator quads of layer 2 (VALU head slices)
  static_assert(U1 >= P1 && U2 >= P2 && U3 >= P3 && U1 > NQ, "chain too small for the pipeline depths");

  // ---- narrow head on the VALU: y[m] += sum_j w3[k(q,j)][m] * act2(acc2 quad q + b2 quad q) ----
  float yacc[C::L3V ? C::D3 : 1];
  auto head_slice = [&](const f32x16(&p2)[C::MT2], int q) {
    const int kt = q / 4, rg = q % 4;
    const f32x4 bq = b2[q * 2];
    f32x4 hv;
#pragma unroll
    for (int j = 0; j < 4; j++) hv[j] = apply_act_c<C::A2>(p2[kt][4 * rg + j] + bq[j], 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < (C::L3V ? C::D3 : 0); m++) {
      const f32x4 wq = w3[q * 2 * C::D3 + m];
#pragma unroll
      for (int j = 0; j < 4; j++) yacc[m] = fmaf(wq[j], hv[j], yacc[m]);
    }
  };
  auto head_store = [&](int64_t t) {
    const int64_t row = (t << 5) + r;
#pragma unroll
    for (int m = 0; m < (C::L3V ? C::D3 : 0); m++) {
      const float tot = yacc[m] + __shfl_xor(yacc[m], 32);  // the two lane halves hold disjoint k
      if (h == 0 && t >= 0 && row < rows) Y[row * C::D3 + m] = apply_act_c<C::A3>(tot + b3[m], 0.f, 0.f);
      yacc[m] = 0.f;
    }
  };

  // layer-2 fragments that fit neither LDS nor the per-tile working set's shadow: loaded ONCE per wave
  // and kept in registers (loop invariant on purpose).
  f32x4 wreg[C::NR2 > 0 ? C::NR2 : 1];
#pragma unroll
  for (int i = 0; i < C::NR2; i++) wreg[i] = w2g_base[(U2L + i) * 64];

  f32x4 x[C::G0];
  load_x(x, tile);
  f32x16 pend[C::MT2];  // layer-2 accumulators of the previous tile, consumed by the VALU head
#pragma unroll
  for (int mt = 0; mt < C::MT2; mt++)
#pragma unroll
    for (int i = 0; i < 16; i++) pend[mt][i] = 0.f;
#pragma unroll
  for (int m = 0; m < (C::L3V ? C::D3 : 1); m++) yacc[m] = 0.f;
  int64_t pend_tile = -1;

  for (; tile < ntiles; tile += tstride) {
    const bool has_next = tile + tstride < ntiles;
    // The fragment addresses are loop invariant and hipcc's LICM would hoist all layer-2 loads out of
    // the tile loop (= W2 held in 256 VGPRs, everything else spilled).  Launder an integer offset per
    // tile (not the pointer itself: that drops it to the flat address space, and flat loads also count
    // on lgkmcnt, i.e. every LDS wait would wait for L2).
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const f32x4 *w2g = w2g_base + zero;
    auto frag2 = [&](int u) -> f32x4 { return u < U2L ? w2l[u * 64] : (u < U2R ? wreg[u - U2L] : w2g[u * 64]); };
    constexpr int LAST_G = U2R < U2 ? U2 - P2 - 1 : -1;  // unit that issues the last L2 fragment load (-1: none)

    // ================= layer 1: acc1[mt] = W1^T . X^T   (+ VALU head of the previous tile) =================
    f32x16 acc1[C::MT1];
#pragma unroll
    for (int mt = 0; mt < C::MT1; mt++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc1[mt][i] = 0.f;
    {
      f32x4 ring1[P1];
#pragma unroll
      for (int u = 0; u < P1; u++) ring1[u] = w1[u * 64];
#pragma unroll
      for (int u = 0; u < U1; u++) {
        const int g = u / C::MT1, mt = u % C::MT1;
        const f32x4 a = ring1[u % P1];
        if (u + P1 < U1) ring1[u % P1] = w1[(u + P1) * 64];
        if constexpr (C::L3V) {
          if (u < NQ) head_slice(pend, u);
          if (u == NQ) head_store(pend_tile);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x[g][j], acc1[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ================= layer 2: acc2[mt] = W2^T . act1(acc1 + b1) =================
    f32x16 acc2[C::MT2];
#pragma unroll
    for (int mt = 0; mt < C::MT2; mt++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc2[mt][i] = 0.f;
    {
      f32x4 ring2[P2], bring1[2];
#pragma unroll
      for (int u = 0; u < P2; u++) ring2[u] = frag2(u);
      bring1[0] = b1[0];
#pragma unroll
      for (int g = 0; g < C::G1; g++) {
        const int kt = g / 4, rg = g % 4;
        const f32x4 bq = bring1[g % 2];
        if (g + 1 < C::G1) bring1[(g + 1) % 2] = b1[(g + 1) * 2];  // bias quad one group ahead
        float hv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) hv[j] = apply_act_c<C::A1>(acc1[kt][4 * rg + j] + bq[j], 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < C::MT2; mt++) {
          const int u = g * C::MT2 + mt;
          const f32x4 a = ring2[u % P2];
          if (u + P2 < U2) ring2[u % P2] = frag2(u + P2);
          // Next tile's X rows (HBM): requested right after the LAST layer-2 L2 load of this tile.
          // vmcnt retires in order, so nothing in this tile waits behind HBM latency, and the request
          // still has the tail of layer 2 (P2 units) to land.  x is dead since the end of layer 1.
          if (u == (LAST_G >= 0 ? LAST_G : 0) && has_next) load_x(x, tile + tstride);
#pragma unroll
          for (int j = 0; j < 4; j++) acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], hv[j], acc2[mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    if constexpr (C::L3V) {
      // hand the layer-2 accumulators to the next iteration's VALU head
#pragma unroll
      for (int mt = 0; mt < C::MT2; mt++) pend[mt] = acc2[mt];
      pend_tile = tile;
    } else {
      // ================= layer 3 on the matrix cores: acc3 = W3^T . act2(acc2 + b2)  (D3 <= 32) =================
      f32x16 acc3;
#pragma unroll
      for (int i = 0; i < 16; i++) acc3[i] = 0.f;
      f32x4 ring3[P3], bring2[2];
      bring2[0] = b2[0];
#pragma unroll
      for (int u = 0; u < P3; u++) ring3[u] = w3[u * 64];
#pragma unroll
      for (int g = 0; g < U3; g++) {
        const int kt = g / 4, rg = g % 4;
        const f32x4 bq = bring2[g % 2];
        if (g + 1 < U3) bring2[(g + 1) % 2] = b2[(g + 1) * 2];
        const f32x4 a = ring3[g % P3];
        if (g + P3 < U3) ring3[g % P3] = w3[(g + P3) * 64];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float hv = apply_act_c<C::A2>(acc2[kt][4 * rg + j] + bq[j], 0.f, 0.f);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], hv, acc3, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: Y[row, f] = act3(acc3 + b3), f = 8*rg + 4h + j < D3
      const int64_t row = (tile << 5) + r;
      if (row < rows) {
        float *yrow = Y + row * C::D3;
        const f32x4 *b3q = reinterpret_cast<const f32x4 *>(b3) + h;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const f32x4 bq = b3q[rg * 2];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int f = 8 * rg + 4 * h + j;  // h is runtime: both halves test f < D3
            if (8 * rg + j < C::D3 && f < C::D3) yrow[f] = apply_act_c<C::A3>(acc3[4 * rg + j] + bq[j], 0.f, 0.f);
          }
        }
      }
    }
  }

  if constexpr (C::L3V) {  // drain: head of the last tile
#pragma unroll
    for (int q = 0; q < NQ; q++) head_slice(pend, q);
    head_store(pend_tile);
  }
}

// ---- two-waves-per-SIMD variant ------------------------------------------------------------------------
// Same math and packed blob as mlp3_kernel, but 8 waves per workgroup (two per SIMD, <= 256 registers
// each) so that one wave's VALU-only phases, s_waitcnt stalls and epilogue are covered by the sibling
// wave's MFMAs.  To fit 256 registers layer 1 is evaluated in SPLIT feature halves; each half is fed to
// layer 2 (a partial sum over that half's k range) before the next half is computed, so only
// MT1/SPLIT accumulator tiles are live at a time.  X stays in registers across the halves.  The narrow
// head runs on the VALU at the end of the tile (no cross-tile software pipelining needed here).
template <class C, int SPLIT, int P2S, int NW>
__global__ __launch_bounds__(NW * 64) void mlp3_split_kernel(const float *__restrict__ X, const float *__restrict__ packed,
                                                        float *__restrict__ Y, int64_t rows) {
  static_assert(C::L3V && C::MT1 % SPLIT == 0 && C::G1 % SPLIT == 0, "split kernel: VALU head, even split");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(packed);
    f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
    for (int i = threadIdx.x; i < C::L_SMALL / 4; i += NW * 64) dst[i] = src[i];
    const f32x4 *ssrc = reinterpret_cast<const f32x4 *>(packed + C::OFF_SMALL);
    f32x4 *sdst = reinterpret_cast<f32x4 *>(lds + C::L_SMALL);
    for (int i = threadIdx.x; i < C::N_SMALL / 4; i += NW * 64) sdst[i] = ssrc[i];
  }
  __syncthreads();
  const f32x4 *w1 = reinterpret_cast<const f32x4 *>(lds + C::L_W1) + lane;
  const f32x4 *w2l = reinterpret_cast<const f32x4 *>(lds + C::L_W2) + lane;
  const float *small = lds + C::L_SMALL;
  const f32x4 *b1 = reinterpret_cast<const f32x4 *>(small + C::S_B1) + h;
  const f32x4 *b2 = reinterpret_cast<const f32x4 *>(small + C::S_B2) + h;
  const f32x4 *w3 = reinterpret_cast<const f32x4 *>(small + C::S_W3) + h * C::D3;
  const float *b3 = small + C::S_B3;
  const f32x4 *w2g_base = reinterpret_cast<const f32x4 *>(packed + C::OFF_W2) + lane;

  const int64_t ntiles = (rows + 31) >> 5;
  const int64_t tstride = int64_t(gridDim.x) * NW;
  int64_t tile = int64_t(blockIdx.x) * NW + wave;
  if (tile >= ntiles) return;

  auto load_x = [&](f32x4(&x)[C::G0], int64_t t) {
    int64_t row = (t << 5) + r;
    if (row >= rows) row = rows - 1;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(X + row * C::D0 + 4 * h);
#pragma unroll
    for (int g = 0; g < C::G0; g++) x[g] = p[2 * g];
  };

  constexpr int MTH = C::MT1 / SPLIT, G1H = C::G1 / SPLIT;
  constexpr int U1H = C::G0 * MTH, P1 = C::P1;
  constexpr int U2 = C::G1 * C::MT2, U2H = G1H * C::MT2, P2 = P2S;
  constexpr int U2L = C::NL2 * C::MT2;
  constexpr int NQ = C::MT2 * 4;
  // unit after which next tile's X may be requested: x must be dead (last split) and, if possible, the
  // last L2 fragment load of the tile already issued
  constexpr int LAST_G = U2L < U2 ? U2 - P2 - 1 : -1;
  constexpr int XPF = LAST_G > U2 - U2H ? LAST_G : U2 - U2H;
  static_assert(U1H >= P1 && U2H >= P2, "chain too small for the pipeline depths");

  f32x4 x[C::G0];
  load_x(x, tile);
  for (; tile < ntiles; tile += tstride) {
    const bool has_next = tile + tstride < ntiles;
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const f32x4 *w2g = w2g_base + zero;
    auto frag2 = [&](int u) -> f32x4 { return u < U2L ? w2l[u * 64] : w2g[u * 64]; };

    f32x16 acc2[C::MT2];
#pragma unroll
    for (int mt = 0; mt < C::MT2; mt++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc2[mt][i] = 0.f;
    f32x4 ring2[P2];
#pragma unroll
    for (int u = 0; u < P2; u++) ring2[u] = frag2(u);

#pragma unroll
    for (int s = 0; s < SPLIT; s++) {
      // ---- layer 1, feature tiles [s*MTH, (s+1)*MTH) ----
      f32x16 acc1[MTH];
#pragma unroll
      for (int mt = 0; mt < MTH; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc1[mt][i] = 0.f;
      {
        f32x4 ring1[P1];
#pragma unroll
        for (int u = 0; u < P1; u++) ring1[u] = w1[((u / MTH) * C::MT1 + s * MTH + (u % MTH)) * 64];
#pragma unroll
        for (int u = 0; u < U1H; u++) {
          const int g = u / MTH, mt = u % MTH;
          const f32x4 a = ring1[u % P1];
          if (u + P1 < U1H) ring1[u % P1] = w1[(((u + P1) / MTH) * C::MT1 + s * MTH + ((u + P1) % MTH)) * 64];
#pragma unroll
          for (int j = 0; j < 4; j++) acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x[g][j], acc1[mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- layer 2, k groups [s*G1H, (s+1)*G1H): partial sums into acc2 ----
      f32x4 bring1[2];
      bring1[0] = b1[(s * G1H) * 2];
#pragma unroll
      for (int gl = 0; gl < G1H; gl++) {
        const int g = s * G1H + gl, ktl = gl / 4, rg = gl % 4;
        const f32x4 bq = bring1[gl % 2];
        if (gl + 1 < G1H) bring1[(gl + 1) % 2] = b1[(g + 1) * 2];
        float hv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) hv[j] = apply_act_c<C::A1>(acc1[ktl][4 * rg + j] + bq[j], 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < C::MT2; mt++) {
          const int u = g * C::MT2 + mt;
          const f32x4 a = ring2[u % P2];
          if (u + P2 < U2) ring2[u % P2] = frag2(u + P2);
          if (u == XPF && has_next) load_x(x, tile + tstride);  // x is dead in the last split
#pragma unroll
          for (int j = 0; j < 4; j++) acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], hv[j], acc2[mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- narrow head on the VALU ----
    float yacc[C::D3];
#pragma unroll
    for (int m = 0; m < C::D3; m++) yacc[m] = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int kt = q / 4, rg = q % 4;
      const f32x4 bq = b2[q * 2];
#pragma unroll
      for (int m = 0; m < C::D3; m++) {
        const f32x4 wq = w3[q * 2 * C::D3 + m];
#pragma unroll
        for (int j = 0; j < 4; j++) yacc[m] = fmaf(wq[j], apply_act_c<C::A2>(acc2[kt][4 * rg + j] + bq[j], 0.f, 0.f), yacc[m]);
      }
    }
    const int64_t row = (tile << 5) + r;
#pragma unroll
    for (int m = 0; m < C::D3; m++) {
      const float tot = yacc[m] + __shfl_xor(yacc[m], 32);
      if (h == 0 && row < rows) Y[row * C::D3 + m] = apply_act_c<C::A3>(tot + b3[m], 0.f, 0.f);
    }
  }
}

template <class C, int SPLIT, int P2S, int NW = 8>
void launch_split(hipStream_t s, const float *X, const float *packed, float *Y, int64_t rows, int num_cus) {
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp3_split_kernel<C, SPLIT, P2S, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::N_LDS * 4);
    attr_done.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
  }
  const int64_t ntiles = (rows + 31) / 32;
  int64_t blocks = (ntiles + NW - 1) / NW;
  if (blocks > num_cus) blocks = num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((mlp3_split_kernel<C, SPLIT, P2S, NW>), dim3((unsigned)blocks), dim3(NW * 64), C::N_LDS * 4, s, X, packed, Y, rows);
}

// ---- host side ----------------------------------------------------------------------------------

// k index consumed by MFMA k-step s (0-based within the layer) on lane half h.
inline int k_of(int s, int h) { return 8 * (s >> 2) + 4 * h + (s & 3); }

// Fragment-major packing of W[K, M] (row-major) for a layer with K % 8 == 0 and MT 32-wide tiles
// (columns >= M are zero).
void pack_frags(const float *W, int K, int M, int MT, float *out) {
  const int G = K / 8;
  for (int g = 0; g < G; g++)
    for (int mt = 0; mt < MT; mt++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 4; j++) {
          const int k = k_of(4 * g + j, lane >> 5), m = 32 * mt + (lane & 31);
          out[((size_t(g) * MT + mt) * 64 + lane) * 4 + j] = m < M ? W[size_t(k) * M + m] : 0.f;
        }
}

// Bias quads [mt][rg][h][4]: feature 32mt + 8rg + 4h + j.
void pack_bias(const float *b, int M, int MT, float *out) {
  for (int mt = 0; mt < MT; mt++)
    for (int rg = 0; rg < 4; rg++)
      for (int h = 0; h < 2; h++)
        for (int j = 0; j < 4; j++) {
          const int f = 32 * mt + 8 * rg + 4 * h + j;
          out[((mt * 4 + rg) * 2 + h) * 4 + j] = (b && f < M) ? b[f] : 0.f;
        }
}

// VALU-head weight quads [q = kt*4+rg][h][m][4]: W3[32kt + 8rg + 4h + j][m].
void pack_head(const float *W3, int K, int D3, float *out) {
  for (int q = 0; q < K / 8; q++)
    for (int h = 0; h < 2; h++)
      for (int m = 0; m < D3; m++)
        for (int j = 0; j < 4; j++) out[((q * 2 + h) * D3 + m) * 4 + j] = W3[size_t(8 * q + 4 * h + j) * D3 + m];
}

template <class C>
void pack_cfg(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3, float *p) {
  pack_frags(W1, C::D0, C::D1, C::MT1, p + C::OFF_W1);
  pack_frags(W2, C::D1, C::D2, C::MT2, p + C::OFF_W2);
  float *s = p + C::OFF_SMALL;
  pack_bias(b1, C::D1, C::MT1, s + C::S_B1);
  pack_bias(b2, C::D2, C::MT2, s + C::S_B2);
  if constexpr (C::L3V) {
    pack_head(W3, C::D2, C::D3, s + C::S_W3);
    for (int m = 0; m < 4; m++) s[C::S_B3 + m] = (b3 && m < C::D3) ? b3[m] : 0.f;
  } else {
    pack_frags(W3, C::D2, C::D3, C::MT3, s + C::S_W3);
    pack_bias(b3, C::D3, C::MT3, s + C::S_B3);
  }
}

template <class C>
void launch_cfg(hipStream_t s, const float *X, const float *packed, float *Y, int64_t rows, int num_cus) {
  // > 64 KiB of dynamic LDS needs the attribute, once per device (not per launch: it is a runtime
  // API call behind a lock, and launches may be inside a stream capture).
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp3_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::N_LDS * 4);
    attr_done.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
  }
  const int64_t ntiles = (rows + 31) / 32;
  int64_t blocks = (ntiles + C::WAVES - 1) / C::WAVES;
  if (blocks > num_cus) blocks = num_cus;  // persistent: one workgroup per CU (the LDS image allows only one)
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((mlp3_kernel<C>), dim3((unsigned)blocks), dim3(C::WAVES * 64), C::N_LDS * 4, s, X, packed, Y, rows);
}

// Ahead-of-time instantiations.  (act codes: 0 none, 1 relu)
using CfgC2 = Cfg<128, 256, 64, 1, 1, 1, 0, 4>;        // BASELINE C2/C3: regression head
using CfgC2x3 = Cfg<128, 256, 64, 3, 1, 1, 0, 4>;      // same trunk, 3 outputs (VALU head)
using CfgC2x10 = Cfg<128, 256, 64, 10, 1, 1, 0, 4>;    // same trunk, 10 outputs (MFMA head)

#define INFERA_MLP3_CONFIGS(X_) X_(CfgC2) X_(CfgC2x3) X_(CfgC2x10)

// Tuning variants of the C2 instantiation, selectable at run time with INFERA_MLP3_VARIANT=<n> for
// within-process A/B runs (tools/ab_mlp3.py, build with `make PROBES=1`).  Variant 0 is the shipped
// kernel.  The packed-blob layout depends only on L3V, so variants are grouped by it.
#ifdef INFERA_MLP3_PROBES
// Round-1 findings (10M rows, ms per launch): P2=8 7.24 | P2=16 7.11 (shipped) | all non-LDS layer-2
// units pinned in registers 7.16-7.23 | NL2=0 (all of W2 from L2) 7.34 | MFMA head instead of VALU 7.66.
using CfgC2_v1 = Cfg<128, 256, 64, 1, 1, 1, 0, 4, 3, 8, -1, 0>;    // shallower layer-2 ring
using CfgC2_v2 = Cfg<128, 256, 64, 1, 1, 1, 0, 4, 3, 24, -1, 0>;   // deeper layer-2 ring
using CfgC2_v3 = Cfg<128, 256, 64, 1, 1, 1, 0, 4, 4, 16, -1, 0>;   // deeper layer-1 ring
using CfgC2_v4 = Cfg<128, 256, 64, 1, 1, 1, 0, 4, 2, 16, -1, 0>;
using CfgC2_v5 = Cfg<128, 256, 64, 1, 1, 1, 0, 4, 3, 3, -1, -1>;   // rest of W2 in registers
#endif

template <class C>
bool matches(const Mlp3Shape &sh) {
  return sh.d0 == C::D0 && sh.d1 == C::D1 && sh.d2 == C::D2 && sh.d3 == C::D3 && sh.act1 == C::A1 && sh.act2 == C::A2 &&
         sh.act3 == C::A3;
}

}  // namespace

bool mlp3_supported(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return true;
  INFERA_MLP3_CONFIGS(X_)
#undef X_
  return false;
}

size_t mlp3_packed_floats(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return size_t(C::N_TOTAL);
  INFERA_MLP3_CONFIGS(X_)
#undef X_
  return 0;
}

void mlp3_pack(const Mlp3Shape &sh, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
               const float *b3, float *packed) {
#define X_(C) if (matches<C>(sh)) { pack_cfg<C>(W1, b1, W2, b2, W3, b3, packed); return; }
  INFERA_MLP3_CONFIGS(X_)
#undef X_
}

void mlp3(hipStream_t s, const Mlp3Shape &sh, const float *X, const float *packed, float *Y, int64_t rows, int num_cus) {
  if (rows <= 0) return;
#ifdef INFERA_MLP3_PROBES
  if (matches<CfgC2>(sh)) {
    const char *v = std::getenv("INFERA_MLP3_VARIANT");
    switch (v ? std::atoi(v) : 0) {
      case 1: launch_cfg<CfgC2_v1>(s, X, packed, Y, rows, num_cus); return;
      case 2: launch_cfg<CfgC2_v2>(s, X, packed, Y, rows, num_cus); return;
      case 3: launch_cfg<CfgC2_v3>(s, X, packed, Y, rows, num_cus); return;
      case 4: launch_cfg<CfgC2_v4>(s, X, packed, Y, rows, num_cus); return;
      case 5: launch_cfg<CfgC2_v5>(s, X, packed, Y, rows, num_cus); return;
      case 6: launch_split<CfgC2, 2, 8>(s, X, packed, Y, rows, num_cus); return;
      case 7: launch_split<CfgC2, 2, 4>(s, X, packed, Y, rows, num_cus); return;
      case 8: launch_split<CfgC2, 4, 8>(s, X, packed, Y, rows, num_cus); return;
      case 9: launch_split<CfgC2, 8, 8>(s, X, packed, Y, rows, num_cus); return;
      case 10: launch_split<CfgC2, 8, 4, 12>(s, X, packed, Y, rows, num_cus); return;
      case 11: launch_split<CfgC2, 4, 4, 12>(s, X, packed, Y, rows, num_cus); return;
      case 12: launch_split<CfgC2, 4, 4>(s, X, packed, Y, rows, num_cus); return;
      case 13: launch_cfg<CfgC2>(s, X, packed, Y, rows, num_cus); return;  // the one-wave-per-SIMD kernel
      default: break;
    }
  }
#endif
  // shipped choice (round-1 A/B, 10M rows): two waves per SIMD with layer 1 in 4 feature slices 6.77-6.83 ms
  // vs one wave per SIMD 7.11 ms; 3 waves per SIMD 6.9 ms.  Heads wider than 4 keep the 1-wave kernel.
#define X_(C)                                                                  \
  if (matches<C>(sh)) {                                                        \
    if constexpr (C::L3V) launch_split<C, 4, 8, 8>(s, X, packed, Y, rows, num_cus); \
    else launch_cfg<C>(s, X, packed, Y, rows, num_cus);                        \
    return;                                                                    \
  }
  INFERA_MLP3_CONFIGS(X_)
#undef X_
}

const char *mlp3_kernel_name(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return C::L3V ? "mlp3_split_kernel<" #C ", 4, 8, 8>" : "mlp3_kernel<" #C ">";
  INFERA_MLP3_CONFIGS(X_)
#undef X_
  return "";
}

}  // namespace infera_hip::kern
