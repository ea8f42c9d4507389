// device_common.hpp -- small device helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.hpp"

namespace infera_hip::kern {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Same formulas as the CPU oracle (oracle/infera_oracle.c op_unary): sigmoid = 1/(1+exp(-x)) etc.
// expf/tanhf are the accurate device-library versions (<= 2 ulp), not the fast intrinsics.
template <int KIND>
__device__ __forceinline__ float apply_act_c(float v, float a, float b) {
  if constexpr (KIND == 1) return v > 0.f ? v : 0.f;
  if constexpr (KIND == 2) return 1.0f / (1.0f + expf(-v));
  if constexpr (KIND == 3) return tanhf(v);
  if constexpr (KIND == 4) return v >= 0.f ? v : a * v;
  if constexpr (KIND == 5) return v < a ? a : (v > b ? b : v);
  if constexpr (KIND == 6) return expf(v);
  if constexpr (KIND == 7) return logf(v);
  if constexpr (KIND == 8) return sqrtf(v);
  if constexpr (KIND == 9) return -v;
  if constexpr (KIND == 10) return fabsf(v);
  if constexpr (KIND == 11) return v >= 0.f ? v : a * (expf(v) - 1.0f);             // Elu
  if constexpr (KIND == 12) return v > 0.f ? b * v : b * (a * expf(v) - a);          // Selu (b = gamma)
  if constexpr (KIND == 13) return logf(expf(v) + 1.0f);                             // Softplus
  if constexpr (KIND == 14) return fmaxf(0.f, fminf(1.f, a * v + b));                // HardSigmoid
  if constexpr (KIND == 15) return v * fmaxf(0.f, fminf(1.f, v * (1.0f / 6.0f) + 0.5f));  // HardSwish
  if constexpr (KIND == 16) return erff(v);
  if constexpr (KIND == 17) return 0.5f * v * (1.0f + erff(v * 0.707106781186547524f));  // Gelu (exact)
  if constexpr (KIND == 18) return 1.0f / v;
  if constexpr (KIND == 19) return floorf(v);
  if constexpr (KIND == 20) return ceilf(v);
  if constexpr (KIND == 21) return v / (1.0f + fabsf(v));                            // Softsign
  if constexpr (KIND == 22) return truncf(v);
  if constexpr (KIND == 23) return rintf(v);                                         // Round: half to even
  if constexpr (KIND == 24) return v / (1.0f + expf(-v));                            // Swish = x * sigmoid(x)
  return v;
}

// Full run-time switch: only for the HBM-bound elementwise kernels (the MFMA epilogues use dispatch_act, and
// the lowering never fuses a kind > 5 into them).
__device__ __forceinline__ float apply_act(float v, const ActParam &p) {
  switch (p.kind) {
#define INFERA_ACT_CASE(K) \
  case K: return apply_act_c<K>(v, p.a, p.b);
    INFERA_ACT_CASE(1) INFERA_ACT_CASE(2) INFERA_ACT_CASE(3) INFERA_ACT_CASE(4) INFERA_ACT_CASE(5) INFERA_ACT_CASE(6)
    INFERA_ACT_CASE(7) INFERA_ACT_CASE(8) INFERA_ACT_CASE(9) INFERA_ACT_CASE(10) INFERA_ACT_CASE(11) INFERA_ACT_CASE(12)
    INFERA_ACT_CASE(13) INFERA_ACT_CASE(14) INFERA_ACT_CASE(15) INFERA_ACT_CASE(16) INFERA_ACT_CASE(17) INFERA_ACT_CASE(18)
    INFERA_ACT_CASE(19) INFERA_ACT_CASE(20) INFERA_ACT_CASE(21) INFERA_ACT_CASE(22) INFERA_ACT_CASE(23) INFERA_ACT_CASE(24)
#undef INFERA_ACT_CASE
    default: return v;
  }
}

// Run f(std::integral_constant<int, KIND>) for the run-time activation kind: ONE wave-uniform branch per
// epilogue instead of a switch per element.  (A per-element switch in an MFMA epilogue costs far more than
// its instructions: each element's bias load gets its own s_waitcnt vmcnt(0), which also drains the stores
// issued just before it -- measured in the conv epilogue at twice the time of its whole main loop.)
template <typename F>
__device__ __forceinline__ void dispatch_act(int kind, F &&f) {
  switch (kind) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 5: f(std::integral_constant<int, 5>{}); break;
    case 14: f(std::integral_constant<int, 14>{}); break;  // HardSigmoid, HardSwish, Swish: the gates of mobile nets
    case 15: f(std::integral_constant<int, 15>{}); break;
    case 24: f(std::integral_constant<int, 24>{}); break;
    default: f(std::integral_constant<int, 0>{}); break;
  }
}

__device__ __forceinline__ float apply_bop(float x, float c, char op, bool const_left) {
  const float l = const_left ? c : x, r = const_left ? x : c;
  switch (op) {
    case '+': return l + r;
    case '-': return l - r;
    case '*': return l * r;
    case '/': return l / r;
    case 'm': return fminf(l, r);
    case 'M': return fmaxf(l, r);
    case '^': return powf(l, r);
    default: return x >= 0.f ? x : c * x;  // 'p': PRelu, c = slope
  }
}

}  // namespace infera_hip::kern
