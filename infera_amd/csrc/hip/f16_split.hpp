// f16_split.hpp -- host-side helpers of the split-fp16 convolution mode (conv_split.hip, the split stem in conv.hip): fp32 <-> fp16 bit
// conversions (round to nearest even, every class of value handled) and the power-of-two scale pair both sides derive from a maximum.
#pragma once

#include <cstdint>
#include <cstring>

namespace infera_hip::kern {

// fp32 -> fp16 bits, round to nearest even
inline uint16_t f16_bits_rne(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return uint16_t(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));  // NaN / inf
  if (x >= 0x477ff000u) return uint16_t(sign | 0x7c00u);                                   // rounds to >= 65520 -> inf
  if (x < 0x33000001u) return uint16_t(sign);                                              // <= 2^-25 -> 0 (ties to even)
  if (x < 0x38800000u) {  // subnormal half: value = m * 2^-24, m = RNE(f * 2^24)
    const int shift = 126 - int(x >> 23);  // 14 <= shift <= 24
    const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
    const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    return uint16_t(sign | (q + ((rem > halfway || (rem == halfway && (q & 1u))) ? 1u : 0u)));
  }
  const uint32_t q = (x - 0x38000000u) >> 13, rem = x & 0x1fffu;  // exponent rebased, 10 mantissa bits kept
  return uint16_t(sign | (q + ((rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ? 1u : 0u)));  // a carry walks into the exponent
}

inline float f16_bits_to_float(uint16_t hbits) {
  const uint32_t sign = uint32_t(hbits & 0x8000u) << 16, e = (hbits >> 10) & 0x1fu, m = hbits & 0x3ffu;
  float f;
  if (e == 0) {  // zero / subnormal: m * 2^-24
    f = float(m) * 5.9604644775390625e-8f;
    return sign ? -f : f;
  }
  const uint32_t x = e == 31 ? (sign | 0x7f800000u | (m << 13)) : (sign | ((e + 112u) << 23) | (m << 13));
  std::memcpy(&f, &x, 4);
  return f;
}

// The scale 2^p that brings magnitudes <= amax into [2^14, 2^15) and its inverse, as float bit patterns: the same arithmetic as the
// kernels' scales_of (exponent clamped so both stay normal floats; amax = 0 or tiny -> the largest scale).
inline void f16_split_scale_bits(float amax, uint32_t &scale_bits, uint32_t &inv_bits) {
  uint32_t bits;
  std::memcpy(&bits, &amax, 4);
  uint32_t e = (bits >> 23) & 0xffu;
  e = e < 15u ? 15u : (e > 254u ? 254u : e);
  scale_bits = (268u - e) << 23;
  inv_bits = (e - 14u) << 23;
}

}  // namespace infera_hip::kern
