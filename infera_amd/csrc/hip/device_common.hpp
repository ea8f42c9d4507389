// device_common.hpp -- small device helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

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
