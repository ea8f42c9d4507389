// mlp_fused.hip -- whole-chain fused MLP  Y = act3(act2(act1(X.W1+b1).W2+b2).W3+b3)  for gfx950.
//
// This is the kernel behind BASELINE config C2/C3 (128 -> 256 -> 64 -> 1 over a 10M-row table):
// 98,432 flop and 516 B of HBM traffic per table row, i.e. bound by the exact-fp32 matrix cores
// (157.3 TFLOP/s -> 1.60e9 rows/s), not by HBM (8 TB/s -> 15.5e9 rows/s).  The design goal is
// therefore: keep v_mfma_f32_32x32x2_f32 issuing back to back, and touch HBM exactly once per
// input byte.
//
// Structure (one persistent workgroup per CU, 4 or 8 waves, no barrier in the main loop):
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
//   * Layer-1 fragments (D0*D1 floats, 128 KiB for C2) + layer-3 fragments + biases live in LDS
//     for the lifetime of the workgroup; layer-2 fragments (64 KiB) stream from L2 -- all CUs read
//     the same 64 KiB, so they stay L2/MALL resident.  X is read with one 16-byte load per lane
//     per 8 k-indices directly in B-fragment shape and prefetched one row-tile ahead.
//   * bias + activation of layer L are applied lazily to each accumulator quad right before it is
//     consumed as a B operand of layer L+1 (VALU work hidden in the MFMA shadow).
#include "device_common.hpp"

namespace infera_hip::kern {

namespace {

template <int D0_, int D1_, int D2_, int D3_, int A1_, int A2_, int A3_, int WAVES_>
struct Cfg {
  static constexpr int D0 = D0_, D1 = D1_, D2 = D2_, D3 = D3_;
  static constexpr int A1 = A1_, A2 = A2_, A3 = A3_, WAVES = WAVES_;
  static_assert(D0 % 8 == 0 && D1 % 32 == 0 && D2 % 32 == 0 && D3 >= 1 && D3 <= 32, "unsupported chain shape");
  static constexpr int G0 = D0 / 8, G1 = D1 / 8, G2 = D2 / 8;      // groups of 4 k-steps per layer input
  static constexpr int MT1 = D1 / 32, MT2 = D2 / 32, MT3 = 1;      // 32-wide output tiles per layer
  // packed blob layout (floats)
  static constexpr int OFF_W1 = 0;
  static constexpr int N_W1 = G0 * MT1 * 256;
  static constexpr int OFF_W3 = OFF_W1 + N_W1;
  static constexpr int N_W3 = G2 * MT3 * 256;
  static constexpr int OFF_B1 = OFF_W3 + N_W3;  // bias quads: [mt][rg][h][4]
  static constexpr int N_B1 = MT1 * 32;
  static constexpr int OFF_B2 = OFF_B1 + N_B1;
  static constexpr int N_B2 = MT2 * 32;
  static constexpr int OFF_B3 = OFF_B2 + N_B2;
  static constexpr int N_B3 = MT3 * 32;
  static constexpr int N_LDS = OFF_B3 + N_B3;   // everything above is LDS resident
  static constexpr int OFF_W2 = N_LDS;          // streamed from L2
  static constexpr int N_W2 = G1 * MT2 * 256;
  static constexpr int N_TOTAL = OFF_W2 + N_W2;
  static_assert(N_LDS * 4 <= 160 * 1024, "LDS-resident part exceeds 160 KiB");
};

template <class C>
__global__ __launch_bounds__(C::WAVES * 64) void mlp3_kernel(const float *__restrict__ X, const float *__restrict__ packed,
                                                            float *__restrict__ Y, int64_t rows) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- stage the LDS-resident weights once per workgroup (coalesced 16 B loads) ----
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(packed);
    f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
    for (int i = threadIdx.x; i < C::N_LDS / 4; i += C::WAVES * 64) dst[i] = src[i];
  }
  __syncthreads();

  const f32x4 *w1 = reinterpret_cast<const f32x4 *>(lds + C::OFF_W1) + lane;
  const f32x4 *w3 = reinterpret_cast<const f32x4 *>(lds + C::OFF_W3) + lane;
  const f32x4 *b1 = reinterpret_cast<const f32x4 *>(lds + C::OFF_B1) + h;
  const f32x4 *b2 = reinterpret_cast<const f32x4 *>(lds + C::OFF_B2) + h;
  const f32x4 *b3 = reinterpret_cast<const f32x4 *>(lds + C::OFF_B3) + h;
  const f32x4 *w2_base = reinterpret_cast<const f32x4 *>(packed + C::OFF_W2) + lane;

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

  // Software pipeline.  The fully unrolled tile body is a stream of "units": one 16-byte weight
  // fragment (A operands of 4 consecutive k-steps for one 32-wide output tile) followed by the 4
  // MFMAs that consume it.  Fragments are fetched P units ahead into a small register ring; a
  // sched_barrier after every unit keeps hipcc from hoisting hundreds of loads to the top of the
  // unrolled body (which spills: 785 VGPRs without it).
  constexpr int U1 = C::G0 * C::MT1, P1 = 3;   // LDS fragments: ~128-cycle latency, unit = 256 cycles
  constexpr int U2 = C::G1 * C::MT2, P2 = 8;   // L2 fragments: prefetch 8 units = 2048 cycles ahead
  constexpr int U3 = C::G2, P3 = 2;
  static_assert(U1 >= P1 && U2 >= P2 && U3 >= P3, "chain too small for the pipeline depths");

  f32x4 x[C::G0];
  load_x(x, tile);

  for (; tile < ntiles; tile += tstride) {
    // first layer-2 fragments: issued before any MFMA of this tile, so they have all of layer 1
    // (>30k cycles) to land.
    const bool has_next = tile + tstride < ntiles;
    // The fragment addresses are loop invariant and hipcc's LICM would hoist all U2 loads out of the
    // tile loop (= W2 held in 256 VGPRs, everything else spilled).  Launder the pointer per tile.
    // (an integer offset, not the pointer itself: laundering the pointer drops it to the flat
    // address space, and flat loads also count on lgkmcnt, i.e. every LDS wait would wait for L2.)
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const f32x4 *w2 = w2_base + zero;
    f32x4 ring2[P2];
#pragma unroll
    for (int u = 0; u < P2; u++) ring2[u] = w2[u * 64];

    // ================= layer 1: acc1[mt] = W1^T . X^T =================
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
    f32x4 bring1[2];
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
        if (u + P2 < U2) ring2[u % P2] = w2[(u + P2) * 64];
        // Next tile's X rows (HBM): issued right after the LAST layer-2 fragment load of this tile.
        // vmcnt retires in order, so a load issued here can only sit in front of next tile's
        // fragment preload -- nothing in this tile waits behind HBM latency -- and it still has the
        // tail of layer 2 plus layer 3 (~4k cycles) to land.  x is dead since the end of layer 1.
        if (u == U2 - P2 && has_next) load_x(x, tile + tstride);
#pragma unroll
        for (int j = 0; j < 4; j++) acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], hv[j], acc2[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ================= layer 3: acc3 = W3^T . act2(acc2 + b2)   (D3 <= 32, one tile) =================
    f32x16 acc3;
#pragma unroll
    for (int i = 0; i < 16; i++) acc3[i] = 0.f;
    {
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
    }

    // ================= epilogue: Y[row, f] = act3(acc3 + b3), f = 8*rg + 4h + j < D3 =================
    const int64_t row = (tile << 5) + r;
    if (row < rows) {
      float *yrow = Y + row * C::D3;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const f32x4 bq = b3[rg * 2];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int f = 8 * rg + 4 * h + j;  // h is runtime: both halves test f < D3
          if (8 * rg + j < C::D3 && f < C::D3) yrow[f] = apply_act_c<C::A3>(acc3[4 * rg + j] + bq[j], 0.f, 0.f);
        }
      }
    }
  }
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

template <class C>
void pack_cfg(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3, float *p) {
  pack_frags(W1, C::D0, C::D1, C::MT1, p + C::OFF_W1);
  pack_frags(W3, C::D2, C::D3, C::MT3, p + C::OFF_W3);
  pack_bias(b1, C::D1, C::MT1, p + C::OFF_B1);
  pack_bias(b2, C::D2, C::MT2, p + C::OFF_B2);
  pack_bias(b3, C::D3, C::MT3, p + C::OFF_B3);
  pack_frags(W2, C::D1, C::D2, C::MT2, p + C::OFF_W2);
}

template <class C>
void launch_cfg(hipStream_t s, const float *X, const float *packed, float *Y, int64_t rows, int num_cus) {
  static bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp3_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::N_LDS * 4);
    return true;
  }();
  (void)attr_set;
  const int64_t ntiles = (rows + 31) / 32;
  int64_t blocks = (ntiles + C::WAVES - 1) / C::WAVES;
  if (blocks > num_cus) blocks = num_cus;  // persistent: one workgroup per CU (LDS footprint allows only one)
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((mlp3_kernel<C>), dim3((unsigned)blocks), dim3(C::WAVES * 64), C::N_LDS * 4, s, X, packed, Y, rows);
}

// Ahead-of-time instantiations.  (act codes: 0 none, 1 relu)
using CfgC2 = Cfg<128, 256, 64, 1, 1, 1, 0, 4>;

#define INFERA_MLP3_CONFIGS(X_) X_(CfgC2)

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
#define X_(C) if (matches<C>(sh)) { launch_cfg<C>(s, X, packed, Y, rows, num_cus); return; }
  INFERA_MLP3_CONFIGS(X_)
#undef X_
}

const char *mlp3_kernel_name(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return "mlp3_kernel<" #C ">";
  INFERA_MLP3_CONFIGS(X_)
#undef X_
  return "";
}

}  // namespace infera_hip::kern
