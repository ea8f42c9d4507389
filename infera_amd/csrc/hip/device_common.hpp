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
  return v;
}

__device__ __forceinline__ float apply_act(float v, const ActParam &p) {
  switch (p.kind) {
    case 1: return apply_act_c<1>(v, p.a, p.b);
    case 2: return apply_act_c<2>(v, p.a, p.b);
    case 3: return apply_act_c<3>(v, p.a, p.b);
    case 4: return apply_act_c<4>(v, p.a, p.b);
    case 5: return apply_act_c<5>(v, p.a, p.b);
    default: return v;
  }
}

// Run f(std::integral_constant<int, KIND>) for the run-time activation kind: ONE wave-uniform branch per
// epilogue instead of a switch per element.  (A per-element switch in an MFMA epilogue costs far more than
// its instructions: each element's bias load gets its own s_waitcnt vmcnt(0), which also drains the stores
// issued just before it -- measured 100k cycles per wave in the conv epilogue, 2x its whole main loop.)
template <typename F>
__device__ __forceinline__ void dispatch_act(int kind, F &&f) {
  switch (kind) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 5: f(std::integral_constant<int, 5>{}); break;
    default: f(std::integral_constant<int, 0>{}); break;
  }
}

__device__ __forceinline__ float apply_bop(float x, float c, char op, bool const_left) {
  const float l = const_left ? c : x, r = const_left ? x : c;
  switch (op) {
    case '+': return l + r;
    case '-': return l - r;
    case '*': return l * r;
    default: return l / r;
  }
}

}  // namespace infera_hip::kern
