// mlp_bf16x3.hip -- OPTIONAL fast mode of the whole-chain fused MLP (INFERA_PRECISION=bf16x3).  NOT the parity path.
//
// The exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate, which is what makes BASELINE
// config C2 MFMA-bound at 1.6e9 rows/s.  Here every fp32 operand is split into two bf16 halves,
// v = hi + lo (hi = RNE_bf16(v), lo = RNE_bf16(v - hi)), and every product is evaluated as
//        a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (the dropped a_lo*b_lo term is <= 2^-16 |a*b|)
// with three v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 3 x 32 cycles per 16 k-values instead of 8 x 64 -- 5.3x less
// matrix time, which moves the chain towards the HBM bound (15.5e9 rows/s).  Per-product relative error ~2^-16..2^-15
// (measured against the oracle in tests/test_bf16x3_gpu.py); it is labelled non-parity precision everywhere it is
// reported and is never the default.
//
// Structure = the fp32 kernel's (mlp_device.inc): everything transposed, H^T[feature, row] = W^T . H_prev^T, table rows
// on the MFMA N axis, so a layer's accumulators ARE the next layer's B operand (after bias + activation + the hi/lo
// split, all on the VALU in the MFMA shadow) -- a 16-k block of layer 2 is {features 32t + 16c + 8(j>>2) + 4h + (j&3)}
// of accumulator tile t, registers 8c..8c+7, and the weight rows are packed in that order.  Persistent, one workgroup per
// CU, ONE wave per SIMD (512 registers per lane): LDS holds all of W1 (hi+lo, 128 KB for C2) and the first layer-2
// blocks; the layer-2 fragments that do not fit stay in REGISTERS for the lifetime of the wave; no barrier after the
// prologue.  The narrow head (D3 <= 4) runs in fp32 on the VALU as in the fp32 kernel.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace infera_hip::kern {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using i64 = long long;

constexpr int kLds = 160 * 1024;

template <int KIND>
__device__ __forceinline__ float act_c(float v) {
  if constexpr (KIND == 1) return fmaxf(v, 0.f);  // one v_max_f32 (the select form costs a canonicalising max more)
  if constexpr (KIND == 2) return 1.0f / (1.0f + expf(-v));
  if constexpr (KIND == 3) return tanhf(v);
  return v;
}

template <int D0_, int D1_, int D2_, int D3_, int A1_, int A2_, int A3_>
struct BCfg {
  static constexpr int D0 = D0_, D1 = D1_, D2 = D2_, D3 = D3_, A1 = A1_, A2 = A2_, A3 = A3_;
  static_assert(D0 % 16 == 0 && D1 % 32 == 0 && D2 % 32 == 0 && D3 >= 1 && D3 <= 4, "unsupported chain shape");
  static constexpr int KB1 = D0 / 16, MT1 = D1 / 32, MT2 = D2 / 32, KB2 = 2 * MT1;
  static constexpr int UNIT = 2048;  // bytes: hi fragment (64 lanes x 16 B) then lo fragment
  static constexpr int N_W1 = KB1 * MT1 * UNIT, N_W2 = KB2 * MT2 * UNIT;
  // small block (floats): bias quads [mt][rg][h][4] of layers 1, 2; head weight quads [q][h][m][4]; head bias (4)
  static constexpr int S_B1 = 0, S_B2 = S_B1 + MT1 * 32, S_W3 = S_B2 + MT2 * 32, S_B3 = S_W3 + MT2 * 32 * D3, N_SMALL = S_B3 + 4;
  static constexpr int OFF_W2 = N_W1, OFF_SMALL = N_W1 + N_W2, N_TOTAL = OFF_SMALL + N_SMALL * 4;  // bytes
  // LDS image: [W1][first NL2 blocks of W2][small]; the other layer-2 units live in registers (8 per unit)
  static constexpr int FIT2 = (kLds - N_W1 - N_SMALL * 4) / (MT2 * UNIT);
  static_assert(FIT2 >= 0, "layer-1 fragments exceed LDS");
  static constexpr int NL2 = FIT2 > KB2 ? KB2 : FIT2;
  static constexpr int NREG = (KB2 - NL2) * MT2;  // register-resident layer-2 units
  static_assert(NREG * 8 <= 200, "too many register-resident layer-2 units");
  static constexpr int L_W2 = N_W1, L_SMALL = N_W1 + NL2 * MT2 * UNIT, N_LDS = L_SMALL + N_SMALL * 4;
};

__device__ __forceinline__ bf16x8 as_bf(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// 8 fp32 values -> bf16 hi and lo fragments (RNE both): v_cvt_pk_bf16_f32 per pair, the hi halves widened back by
// shift / mask for the subtraction
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 &hi, u32x4 &lo) {
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const f32x2 pair = {v[2 * p], v[2 * p + 1]};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(pair, bf16x2));
    const float h0 = __uint_as_float(h << 16), h1 = __uint_as_float(h & 0xffff0000u);
    const f32x2 rest = {v[2 * p] - h0, v[2 * p + 1] - h1};
    hi[p] = h;
    lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(rest, bf16x2));
  }
}

// two fp32 values -> one dword of bf16 hi halves and one of bf16 lo halves
struct HiLo {
  unsigned hi, lo;
};
__device__ __forceinline__ HiLo split_pair(float v0, float v1) {
  const f32x2 pair = {v0, v1};
  const unsigned hbits = __builtin_bit_cast(unsigned, __builtin_convertvector(pair, bf16x2));
  const f32x2 rest = {v0 - __uint_as_float(hbits << 16), v1 - __uint_as_float(hbits & 0xffff0000u)};
  return HiLo{hbits, __builtin_bit_cast(unsigned, __builtin_convertvector(rest, bf16x2))};
}

// one MFMA, then up to `valu` vector-ALU instructions in its 32-cycle shadow (a lone wave issues in order: without this
// the scheduler puts a unit's VALU work in front of its three MFMAs and the matrix pipe waits for it)
#define INFERA_MFMA_THEN_VALU(valu)                    \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
  __builtin_amdgcn_sched_group_barrier(0x002, valu, 0)

template <class C>
__global__ __launch_bounds__(256) void mlp3_bf16x3_kernel(const float *__restrict__ X, const unsigned char *__restrict__ packed,
                                                          float *__restrict__ Y, i64 rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(packed);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    for (int i = threadIdx.x; i < C::L_SMALL / 16; i += 256) dst[i] = src[i];  // W1 + first NL2 blocks of W2 (contiguous)
    const u32x4 *ssrc = reinterpret_cast<const u32x4 *>(packed + C::OFF_SMALL);
    u32x4 *sdst = reinterpret_cast<u32x4 *>(lds + C::L_SMALL);
    for (int i = threadIdx.x; i < C::N_SMALL / 4; i += 256) sdst[i] = ssrc[i];
  }
  __syncthreads();
  const u32x4 *w1 = reinterpret_cast<const u32x4 *>(lds) + lane;           // unit u: hi at w1[u*128], lo at w1[u*128 + 64]
  const u32x4 *w2 = reinterpret_cast<const u32x4 *>(lds + C::L_W2) + lane;
  const float *small = reinterpret_cast<const float *>(lds + C::L_SMALL);
  const f32x4 *b1 = reinterpret_cast<const f32x4 *>(small + C::S_B1) + h;
  const f32x4 *b2 = reinterpret_cast<const f32x4 *>(small + C::S_B2) + h;
  const f32x4 *w3 = reinterpret_cast<const f32x4 *>(small + C::S_W3) + h * C::D3;
  const float *b3 = small + C::S_B3;

  const i64 ntiles = (rows + 31) >> 5;
  const i64 tstride = i64(gridDim.x) * 4;
  i64 tile = i64(blockIdx.x) * 4 + wave;
  if (tile >= ntiles) return;

  // layer-2 units that do not fit in LDS: resident in registers for the lifetime of the wave
  u32x4 rhi[C::NREG > 0 ? C::NREG : 1], rlo[C::NREG > 0 ? C::NREG : 1];
  {
    const u32x4 *g = reinterpret_cast<const u32x4 *>(packed + C::OFF_W2 + C::NL2 * C::MT2 * C::UNIT) + lane;
#pragma unroll
    for (int u = 0; u < C::NREG; u++) {
      rhi[u] = g[u * 128];
      rlo[u] = g[u * 128 + 64];
    }
  }

  constexpr int NX = C::D0 / 4;  // f32x4 pieces of a row held by the two lane halves together: lane half h owns 8 of every 16 columns
  // blocks [b0, b1) of tile t's rows (the first half is requested under the previous tile's layer 2, the second half at
  // the start of layer 1 -- four blocks = 3072 MFMA cycles before its first use: 64 registers less across layer 2)
  auto load_x = [&](f32x4(&x)[C::KB1 * 2], i64 t, int b0, int b1) {
    i64 row = (t << 5) + r;
    if (row >= rows) row = rows - 1;  // tail rows recompute the last row; their stores are masked
    const f32x4 *p = reinterpret_cast<const f32x4 *>(X + row * C::D0 + 8 * h);
#pragma unroll
    for (int b = 0; b < C::KB1; b++)
      if (b >= b0 && b < b1) {
        x[2 * b] = p[4 * b];
        x[2 * b + 1] = p[4 * b + 1];
      }
  };
  constexpr int XH = C::KB1 / 2;
  (void)NX;

  constexpr int P = 3;                          // A-unit ring depth (LDS latency ~128 cycles, a unit = 96 MFMA cycles)
  constexpr int U1 = C::KB1 * C::MT1;           // layer-1 units: one (hi, lo) A pair + 3 MFMAs
  constexpr int U2 = C::KB2 * C::MT2, U2L = C::NL2 * C::MT2;  // layer-2 units; the first U2L come from LDS
  static_assert(U1 >= P && (U2L == 0 || U2L >= P) && C::MT1 >= 2, "chain too small for the pipeline");

  auto split_x = [&](const f32x4(&x)[C::KB1 * 2], int b, u32x4 &hi, u32x4 &lo) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      v[j] = x[2 * b][j];
      v[4 + j] = x[2 * b + 1][j];
    }
    split8(v, hi, lo);
  };

  f32x4 x[C::KB1 * 2];
  load_x(x, tile, 0, XH);
  for (; tile < ntiles; tile += tstride) {
    load_x(x, tile, XH, C::KB1);
    const bool has_next = tile + tstride < ntiles;
    // the fragment offsets are laundered once per tile: otherwise LICM hoists every LDS load of the unrolled body out of
    // the tile loop and spills hundreds of registers (the same measure as in mlp_device.inc)
    // One opaque base per 64 KB window of the LDS image: a ds_read offset is 16 bits, and with a single base the compiler
    // spends one v_add_u32 per fragment read beyond the first window (two per unit = 15 % of the tile's VALU work; measured
    // neutral in time -- the kernel runs at the clock the chip grants it, see the header of DESIGN.md 3.1b).
    constexpr int WIN = 65536 / 16;  // u32x4 elements per window
    const u32x4 *wbase[3] = {w1, w1 + WIN, w1 + 2 * WIN};
#pragma unroll
    for (int k = 0; k < 3; k++) asm volatile("" : "+v"(wbase[k]));
    // unit u of layer 1 / LDS unit u of layer 2, half 0 = hi fragment, 1 = lo fragment
    auto lds_frag = [&](int byte_off) -> u32x4 { return wbase[byte_off / 65536][(byte_off % 65536) / 16]; };
    auto w1f = [&](int u, int half) { return lds_frag(u * C::UNIT + half * 1024); };
    auto w2f = [&](int u, int half) { return lds_frag(C::L_W2 + u * C::UNIT + half * 1024); };
    // ================= layer 1: acc1[t] = b1 + W1^T . X^T, 16 columns per block =================
    // (the bias is the accumulators' initial value: no bias add on the way to layer 2)
    f32x16 acc1[C::MT1];
#pragma unroll
    for (int t = 0; t < C::MT1; t++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 bq = b1[(t * 4 + q) * 2];
#pragma unroll
        for (int j = 0; j < 4; j++) acc1[t][4 * q + j] = bq[j];
      }
    {
      u32x4 rh[P], rl[P];
#pragma unroll
      for (int u = 0; u < P; u++) {
        rh[u] = w1f(u, 0);
        rl[u] = w1f(u, 1);
      }
      u32x4 bh, bl, nbh, nbl;
      split_x(x, 0, bh, bl);
#pragma unroll
      for (int u = 0; u < U1; u++) {
        const int b = u / C::MT1, t = u % C::MT1;
        const u32x4 ah = rh[u % P], al = rl[u % P];
        if (u + P < U1) {
          rh[u % P] = w1f(u + P, 0);
          rl[u % P] = w1f(u + P, 1);
        }
        // the next block's operand, one pair per unit, in this block's MFMA shadow
        if (b + 1 < C::KB1 && t < 4) {
          const f32x4 xv = x[2 * (b + 1) + (t >> 1)];
          const HiLo s2 = split_pair(xv[2 * (t & 1)], xv[2 * (t & 1) + 1]);
          nbh[t] = s2.hi;
          nbl[t] = s2.lo;
        }
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ah), as_bf(bh), acc1[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ah), as_bf(bl), acc1[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(al), as_bf(bh), acc1[t], 0, 0, 0);
        if (t == C::MT1 - 1) {
          bh = nbh;
          bl = nbl;
        }
        INFERA_MFMA_THEN_VALU(3);
        INFERA_MFMA_THEN_VALU(3);
        INFERA_MFMA_THEN_VALU(3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (has_next) load_x(x, tile + tstride, 0, XH);  // x is dead: the first half of the next tile's rows arrives under layer 2
    // ================= layer 2: acc2[t2] = b2 + sum_blocks W2^T[block] . act(acc1)[block] =================
    f32x16 acc2[C::MT2];
#pragma unroll
    for (int t = 0; t < C::MT2; t++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 bq = b2[(t * 4 + q) * 2];
#pragma unroll
        for (int j = 0; j < 4; j++) acc2[t][4 * q + j] = bq[j];
      }
    {
      // pair p (0..3) of block kb = registers 8c + 2p, 8c + 2p + 1 of accumulator tile t
      auto operand_pair = [&](int kb, int p, u32x4 &hi, u32x4 &lo) {
        const int t = kb >> 1, c = kb & 1;
        const HiLo s2 = split_pair(act_c<C::A1>(acc1[t][8 * c + 2 * p]), act_c<C::A1>(acc1[t][8 * c + 2 * p + 1]));
        hi[p] = s2.hi;
        lo[p] = s2.lo;
      };
      u32x4 rh[P], rl[P];
#pragma unroll
      for (int u = 0; u < P && u < U2L; u++) {
        rh[u] = w2f(u, 0);
        rl[u] = w2f(u, 1);
      }
      u32x4 bh, bl, nbh, nbl;
#pragma unroll
      for (int p = 0; p < 4; p++) operand_pair(0, p, bh, bl);
#pragma unroll
      for (int u = 0; u < U2; u++) {
        const int kb = u / C::MT2, t2 = u % C::MT2;
        u32x4 ah, al;
        if (u < U2L) {
          ah = rh[u % P];
          al = rl[u % P];
          if (u + P < U2L) {
            rh[u % P] = w2f(u + P, 0);
            rl[u % P] = w2f(u + P, 1);
          }
        } else {
          ah = rhi[u >= U2L ? u - U2L : 0];
          al = rlo[u >= U2L ? u - U2L : 0];
        }
        if (kb + 1 < C::KB2) {  // the next block's operand, spread over this block's units
          constexpr int PP = (4 + C::MT2 - 1) / C::MT2;  // pairs per unit
#pragma unroll
          for (int p = t2 * PP; p < (t2 + 1) * PP && p < 4; p++) operand_pair(kb + 1, p, nbh, nbl);
        }
        acc2[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ah), as_bf(bh), acc2[t2], 0, 0, 0);
        acc2[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ah), as_bf(bl), acc2[t2], 0, 0, 0);
        acc2[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(al), as_bf(bh), acc2[t2], 0, 0, 0);
        if (t2 == C::MT2 - 1) {
          bh = nbh;
          bl = nbl;
        }
        INFERA_MFMA_THEN_VALU(6);
        INFERA_MFMA_THEN_VALU(6);
        INFERA_MFMA_THEN_VALU(6);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ================= narrow head in fp32 on the VALU (as in mlp_device.inc) =================
    float yacc[C::D3];
#pragma unroll
    for (int m = 0; m < C::D3; m++) yacc[m] = 0.f;
#pragma unroll
    for (int q = 0; q < C::MT2 * 4; q++) {
      const int kt = q / 4, rg = q % 4;
#pragma unroll
      for (int m = 0; m < C::D3; m++) {
        const f32x4 wq = w3[q * 2 * C::D3 + m];
#pragma unroll
        for (int j = 0; j < 4; j++) yacc[m] = fmaf(wq[j], act_c<C::A2>(acc2[kt][4 * rg + j]), yacc[m]);
      }
    }
    const i64 row = (tile << 5) + r;
#pragma unroll
    for (int m = 0; m < C::D3; m++) {
      const float tot = yacc[m] + __shfl_xor(yacc[m], 32);
      if (h == 0 && row < rows) Y[row * C::D3 + m] = act_c<C::A3>(tot + b3[m]);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
uint16_t bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return uint16_t(u >> 16);  // inf / nan
  u += 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(u >> 16);
}
float bf16_to_f32(uint16_t b) {
  const uint32_t u = uint32_t(b) << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
void split_host(float w, uint16_t &hi, uint16_t &lo) {
  hi = bf16_rne(w);
  lo = bf16_rne(w - bf16_to_f32(hi));
}

using CfgB2 = BCfg<128, 256, 64, 1, 1, 1, 0>;    // BASELINE C2 / C3
using CfgB2x3 = BCfg<128, 256, 64, 3, 1, 1, 0>;  // same trunk, 3 outputs
#define INFERA_BF16X3_CONFIGS(X_) X_(CfgB2) X_(CfgB2x3)

template <class C>
bool matches(const Mlp3Shape &sh) {
  return sh.d0 == C::D0 && sh.d1 == C::D1 && sh.d2 == C::D2 && sh.d3 == C::D3 && sh.act1 == C::A1 && sh.act2 == C::A2 && sh.act3 == C::A3;
}

template <class C>
void pack_cfg(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3, unsigned char *out) {
  std::memset(out, 0, size_t(C::N_TOTAL));
  auto put = [&](unsigned char *unit, int lane, int j, float w) {
    uint16_t hi, lo;
    split_host(w, hi, lo);
    std::memcpy(unit + lane * 16 + j * 2, &hi, 2);
    std::memcpy(unit + 1024 + lane * 16 + j * 2, &lo, 2);
  };
  // layer 1: unit (b, t): lane l holds W1[16b + 8(l>>5) + j][32t + (l&31)], j = 0..7
  for (int b = 0; b < C::KB1; b++)
    for (int t = 0; t < C::MT1; t++)
      for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++)
          put(out + size_t(b * C::MT1 + t) * C::UNIT, l, j, W1[size_t(16 * b + 8 * (l >> 5) + j) * C::D1 + 32 * t + (l & 31)]);
  // layer 2: block kb = 2t + c, k-slot (h, j) = feature 32t + 16c + 8(j>>2) + 4h + (j&3) of layer 1's output
  for (int kb = 0; kb < C::KB2; kb++)
    for (int t2 = 0; t2 < C::MT2; t2++)
      for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) {
          const int t = kb >> 1, c = kb & 1, hh = l >> 5, f = 32 * t + 16 * c + 8 * (j >> 2) + 4 * hh + (j & 3);
          put(out + C::OFF_W2 + size_t(kb * C::MT2 + t2) * C::UNIT, l, j, W2[size_t(f) * C::D2 + 32 * t2 + (l & 31)]);
        }
  float *s = reinterpret_cast<float *>(out + C::OFF_SMALL);
  auto bias_quads = [&](const float *b, int M, int MT, float *dst) {  // [mt][rg][h][4]: feature 32mt + 8rg + 4h + j
    for (int mt = 0; mt < MT; mt++)
      for (int rg = 0; rg < 4; rg++)
        for (int hh = 0; hh < 2; hh++)
          for (int j = 0; j < 4; j++) {
            const int f = 32 * mt + 8 * rg + 4 * hh + j;
            dst[((mt * 4 + rg) * 2 + hh) * 4 + j] = (b && f < M) ? b[f] : 0.f;
          }
  };
  bias_quads(b1, C::D1, C::MT1, s + C::S_B1);
  bias_quads(b2, C::D2, C::MT2, s + C::S_B2);
  for (int q = 0; q < C::D2 / 8; q++)  // head weight quads [q][h][m][4]: W3[8q + 4h + j][m]
    for (int hh = 0; hh < 2; hh++)
      for (int m = 0; m < C::D3; m++)
        for (int j = 0; j < 4; j++) s[C::S_W3 + ((q * 2 + hh) * C::D3 + m) * 4 + j] = W3[size_t(8 * q + 4 * hh + j) * C::D3 + m];
  for (int m = 0; m < 4; m++) s[C::S_B3 + m] = (b3 && m < C::D3) ? b3[m] : 0.f;
}

template <class C>
void launch_cfg(hipStream_t s, const float *X, const void *packed, float *Y, int64_t rows, int num_cus) {
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp3_bf16x3_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::N_LDS);
    attr_done.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
  }
  const int64_t ntiles = (rows + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  if (blocks > num_cus) blocks = num_cus;  // persistent: one workgroup per CU
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((mlp3_bf16x3_kernel<C>), dim3(unsigned(blocks)), dim3(256), C::N_LDS, s, X, static_cast<const unsigned char *>(packed), Y, rows);
}

}  // namespace

bool mlp3_bf16x3_supported(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return true;
  INFERA_BF16X3_CONFIGS(X_)
#undef X_
  return false;
}

size_t mlp3_bf16x3_packed_bytes(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return size_t(C::N_TOTAL);
  INFERA_BF16X3_CONFIGS(X_)
#undef X_
  return 0;
}

void mlp3_bf16x3_pack(const Mlp3Shape &sh, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                      const float *b3, void *packed) {
#define X_(C) if (matches<C>(sh)) return pack_cfg<C>(W1, b1, W2, b2, W3, b3, static_cast<unsigned char *>(packed));
  INFERA_BF16X3_CONFIGS(X_)
#undef X_
}

bool mlp3_bf16x3(hipStream_t s, const Mlp3Shape &sh, const float *X, const void *packed, float *Y, int64_t rows, int num_cus) {
  if (rows <= 0) return true;
#define X_(C) if (matches<C>(sh)) { launch_cfg<C>(s, X, packed, Y, rows, num_cus); return true; }
  INFERA_BF16X3_CONFIGS(X_)
#undef X_
  return false;
}

std::string mlp3_bf16x3_kernel_name(const Mlp3Shape &sh) {
#define X_(C) if (matches<C>(sh)) return "mlp3_bf16x3_kernel<" #C ">";
  INFERA_BF16X3_CONFIGS(X_)
#undef X_
  return "";
}

}  // namespace infera_hip::kern
