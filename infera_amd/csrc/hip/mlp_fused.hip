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

#include "../host/common.hpp"
#include <cstdint>
#include <cstdlib>

#include "device_common.hpp"
#include "mlp_device.inc"
#include "mlp_jit.hpp"
#include "mlp_layout.hpp"

namespace infera_hip::kern {

namespace {

using namespace mlpdev;

template <class C, int SPLIT, int P2S, int NW = 8, bool XCM = false>
void launch_split(hipStream_t s, const float *X, const float *packed, float *Y, int64_t rows, int num_cus) {
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp3_split_kernel<C, SPLIT, P2S, NW, XCM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::N_LDS * 4);
    attr_done.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
  }
  const int64_t ntiles = (rows + 31) / 32;
  int64_t blocks = (ntiles + NW - 1) / NW;
  if (blocks > num_cus) blocks = num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((mlp3_split_kernel<C, SPLIT, P2S, NW, XCM>), dim3((unsigned)blocks), dim3(NW * 64), C::N_LDS * 4, s, X, packed, Y, rows);
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

void pack_layout(const Mlp3Layout &L, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                 const float *b3, float *p) {
  pack_frags(W1, L.d0, L.d1, L.MT1, p + L.OFF_W1);
  pack_frags(W2, L.d1, L.d2, L.MT2, p + L.OFF_W2);
  float *s = p + L.OFF_SMALL;
  pack_bias(b1, L.d1, L.MT1, s + L.S_B1);
  pack_bias(b2, L.d2, L.MT2, s + L.S_B2);
  if (L.l3v) {
    pack_head(W3, L.d2, L.d3, s + L.S_W3);
    for (int m = 0; m < 4; m++) s[L.S_B3 + m] = (b3 && m < L.d3) ? b3[m] : 0.f;
  } else {
    pack_frags(W3, L.d2, L.d3, 1, s + L.S_W3);
    pack_bias(b3, L.d3, 1, s + L.S_B3);
  }
}

// The run-time layout (mlp_layout.hpp) must be the compile-time one (mlpdev::Cfg) for every AOT config.
template <class C>
bool layout_matches_cfg() {
  const Mlp3Layout L = mlp3_layout(C::D0, C::D1, C::D2, C::D3);
  return L.l3v == C::L3V && L.N_W1 == C::N_W1 && L.OFF_W2 == C::OFF_W2 && L.OFF_SMALL == C::OFF_SMALL && L.S_B2 == C::S_B2 &&
         L.S_W3 == C::S_W3 && L.S_B3 == C::S_B3 && L.N_TOTAL == C::N_TOTAL && L.NL2 == C::NL2 && L.L_SMALL == C::L_SMALL &&
         L.N_LDS == C::N_LDS;
}

template <class C, bool XCM = false>
void launch_cfg(hipStream_t s, const float *X, const float *packed, float *Y, int64_t rows, int num_cus) {
  // > 64 KiB of dynamic LDS needs the attribute, once per device (not per launch: it is a runtime
  // API call behind a lock, and launches may be inside a stream capture).
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp3_kernel<C, XCM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::N_LDS * 4);
    attr_done.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
  }
  const int64_t ntiles = (rows + 31) / 32;
  int64_t blocks = (ntiles + C::WAVES - 1) / C::WAVES;
  if (blocks > num_cus) blocks = num_cus;  // persistent: one workgroup per CU (the LDS image allows only one)
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((mlp3_kernel<C, XCM>), dim3((unsigned)blocks), dim3(C::WAVES * 64), C::N_LDS * 4, s, X, packed, Y, rows);
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

namespace {
bool aot_match(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return true;
  INFERA_MLP3_CONFIGS(X_)
#undef X_
  return false;
}
}  // namespace

bool mlp3_supported(const Mlp3Shape &sh, std::string *why) {
  if (aot_match(sh)) {
    bool ok = true;
#define X_(C) if (matches<C>(sh)) ok = layout_matches_cfg<C>();
    INFERA_MLP3_CONFIGS(X_)
#undef X_
    if (!ok && why) *why = "internal: run-time layout disagrees with the compiled kernel";
    return ok;
  }
  return mlp3_jit_prepare(sh, why);  // compiles (once) with hipRTC; false -> caller keeps the per-layer plan
}

size_t mlp3_packed_floats(const Mlp3Shape &sh) { return size_t(mlp3_layout(sh.d0, sh.d1, sh.d2, sh.d3).N_TOTAL); }

void mlp3_pack(const Mlp3Shape &sh, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
               const float *b3, float *packed) {
  pack_layout(mlp3_layout(sh.d0, sh.d1, sh.d2, sh.d3), W1, b1, W2, b2, W3, b3, packed);
}

bool mlp3_colmajor_supported(const Mlp3Shape &sh) { return mlp3_colmajor_max_rows(sh) > 0; }
// ahead-of-time configurations read column-major chunks of any length; load-time specialised chains on their tile kernel only
int64_t mlp3_colmajor_max_rows(const Mlp3Shape &sh) { return aot_match(sh) ? INT64_MAX : mlp3_jit_colmajor_max_rows(sh); }

namespace {
// Short launches (a DataChunk through the host ABI): one workgroup per 32-row tile, bit-identical to the split kernel
// (mlp_device.inc, mlp3_tile_kernel).  Up to this many rows the persistent kernels leave most of the chip idle behind their
// 160 KB LDS image: 2048 rows 50 us there, ~12 us here.
constexpr int64_t kTileKernelMaxRows = 32768;
// ... and the shortest ones (a DataChunk is 2048 rows) as 16-row tiles on the 16x16x4 instruction (mlp3_tile16_kernel): twice the
// workgroups, half the layer-1 chain, a quarter of the layer-2 time.  INFERA_MLP_TILE16_MAX_ROWS (measurement knob; default below; 0: never).
template <class C, bool XCM>
void launch_tile(hipStream_t s, const float *X, const float *packed, float *Y, int64_t rows) {
  if constexpr (C::D1 % 64 == 0 && C::D2 <= 128) {
    if (rows <= mlp3_tile16_max_rows()) {
      hipLaunchKernelGGL((mlp3_tile16_kernel<C, XCM>), dim3(unsigned((rows + 15) / 16)), dim3(256), 0, s, X, packed, Y, rows);
      return;
    }
  }
  const int64_t ntiles = (rows + 31) / 32;
  hipLaunchKernelGGL((mlp3_tile_kernel<C, XCM>), dim3(unsigned(ntiles)), dim3(256), 0, s, X, packed, Y, rows);
}
bool tile_kernel_enabled() { return true; }
}  // namespace

bool mlp3(hipStream_t s, const Mlp3Shape &sh, const float *X, const float *packed, float *Y, int64_t rows, int num_cus,
          std::string *why, bool x_colmajor) {
  if (rows <= 0) return true;
  if (x_colmajor) {  // the host path's column-major chunks (ahead-of-time configurations only)
#define X_(C)                                                                        \
  if (matches<C>(sh)) {                                                              \
    if constexpr (C::L3V) {                                                          \
      if (rows <= kTileKernelMaxRows && tile_kernel_enabled()) launch_tile<C, true>(s, X, packed, Y, rows); \
      else launch_split<C, 4, 8, 8, true>(s, X, packed, Y, rows, num_cus);           \
    } else launch_cfg<C, true>(s, X, packed, Y, rows, num_cus);                      \
    return true;                                                                     \
  }
    INFERA_MLP3_CONFIGS(X_)
#undef X_
    return mlp3_jit_launch(s, sh, X, packed, Y, rows, num_cus, why, true);
  }
#ifdef INFERA_MLP3_PROBES
  if (matches<CfgC2>(sh)) {
    const char *v = std::getenv("INFERA_MLP3_VARIANT");
    switch (v ? std::atoi(v) : 0) {
      case 1: launch_cfg<CfgC2_v1>(s, X, packed, Y, rows, num_cus); return true;
      case 2: launch_cfg<CfgC2_v2>(s, X, packed, Y, rows, num_cus); return true;
      case 3: launch_cfg<CfgC2_v3>(s, X, packed, Y, rows, num_cus); return true;
      case 4: launch_cfg<CfgC2_v4>(s, X, packed, Y, rows, num_cus); return true;
      case 5: launch_cfg<CfgC2_v5>(s, X, packed, Y, rows, num_cus); return true;
      case 6: launch_split<CfgC2, 2, 8>(s, X, packed, Y, rows, num_cus); return true;
      case 7: launch_split<CfgC2, 2, 4>(s, X, packed, Y, rows, num_cus); return true;
      case 8: launch_split<CfgC2, 4, 8>(s, X, packed, Y, rows, num_cus); return true;
      case 9: launch_split<CfgC2, 8, 8>(s, X, packed, Y, rows, num_cus); return true;
      case 10: launch_split<CfgC2, 8, 4, 12>(s, X, packed, Y, rows, num_cus); return true;
      case 11: launch_split<CfgC2, 4, 4, 12>(s, X, packed, Y, rows, num_cus); return true;
      case 12: launch_split<CfgC2, 4, 4>(s, X, packed, Y, rows, num_cus); return true;
      case 13: launch_cfg<CfgC2>(s, X, packed, Y, rows, num_cus); return true;  // the one-wave-per-SIMD kernel
      default: break;
    }
  }
#endif
  // shipped choice (round-1 A/B, 10M rows): two waves per SIMD with layer 1 in 4 feature slices 6.77-6.83 ms
  // vs one wave per SIMD 7.11 ms; 3 waves per SIMD 6.9 ms.  Heads wider than 4 keep the 1-wave kernel.
#define X_(C)                                                                  \
  if (matches<C>(sh)) {                                                        \
    if constexpr (C::L3V) {                                                    \
      if (rows <= kTileKernelMaxRows && tile_kernel_enabled()) launch_tile<C, false>(s, X, packed, Y, rows); \
      else launch_split<C, 4, 8, 8>(s, X, packed, Y, rows, num_cus);           \
    } else launch_cfg<C>(s, X, packed, Y, rows, num_cus);                      \
    return true;                                                               \
  }
  INFERA_MLP3_CONFIGS(X_)
#undef X_
  return mlp3_jit_launch(s, sh, X, packed, Y, rows, num_cus, why);
}

std::string mlp3_kernel_name(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return C::L3V ? "mlp3_split_kernel<" #C ", 4, 8, 8>" : "mlp3_kernel<" #C ">";
  INFERA_MLP3_CONFIGS(X_)
#undef X_
  return mlp3_jit_kernel_name(sh);
}

}  // namespace infera_hip::kern
