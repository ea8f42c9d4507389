#include "gather.hpp"

#include <immintrin.h>

namespace infera_hip {

namespace {

template <typename T>
inline float load_as_float(const void *base, size_t i) {
  return static_cast<float>(static_cast<const T *>(base)[i]);
}

inline float cell(const infera::InferaColumn &c, size_t row) {
  const size_t r = c.is_constant ? 0 : row;
  switch (c.type) {
    case infera::INFERA_COL_FLOAT: return load_as_float<float>(c.data, r);
    case infera::INFERA_COL_DOUBLE: return load_as_float<double>(c.data, r);
    case infera::INFERA_COL_INTEGER: return load_as_float<int32_t>(c.data, r);
    default: return load_as_float<int64_t>(c.data, r);
  }
}

// 8 rows x 8 columns: in[j] = 8 consecutive rows of column j; out row i gets columns c0..c0+7.
__attribute__((target("avx2"))) inline void transpose8x8(const float *const in[8], float *dst, size_t ld) {
  __m256 r0 = _mm256_loadu_ps(in[0]), r1 = _mm256_loadu_ps(in[1]), r2 = _mm256_loadu_ps(in[2]), r3 = _mm256_loadu_ps(in[3]);
  __m256 r4 = _mm256_loadu_ps(in[4]), r5 = _mm256_loadu_ps(in[5]), r6 = _mm256_loadu_ps(in[6]), r7 = _mm256_loadu_ps(in[7]);
  __m256 t0 = _mm256_unpacklo_ps(r0, r1), t1 = _mm256_unpackhi_ps(r0, r1), t2 = _mm256_unpacklo_ps(r2, r3), t3 = _mm256_unpackhi_ps(r2, r3);
  __m256 t4 = _mm256_unpacklo_ps(r4, r5), t5 = _mm256_unpackhi_ps(r4, r5), t6 = _mm256_unpacklo_ps(r6, r7), t7 = _mm256_unpackhi_ps(r6, r7);
  __m256 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xEE), u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xEE);
  __m256 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xEE), u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xEE);
  _mm256_storeu_ps(dst + 0 * ld, _mm256_permute2f128_ps(u0, u4, 0x20));
  _mm256_storeu_ps(dst + 1 * ld, _mm256_permute2f128_ps(u1, u5, 0x20));
  _mm256_storeu_ps(dst + 2 * ld, _mm256_permute2f128_ps(u2, u6, 0x20));
  _mm256_storeu_ps(dst + 3 * ld, _mm256_permute2f128_ps(u3, u7, 0x20));
  _mm256_storeu_ps(dst + 4 * ld, _mm256_permute2f128_ps(u0, u4, 0x31));
  _mm256_storeu_ps(dst + 5 * ld, _mm256_permute2f128_ps(u1, u5, 0x31));
  _mm256_storeu_ps(dst + 6 * ld, _mm256_permute2f128_ps(u2, u6, 0x31));
  _mm256_storeu_ps(dst + 7 * ld, _mm256_permute2f128_ps(u3, u7, 0x31));
}

__attribute__((target("avx2"))) void gather_float_avx2(const infera::InferaColumn *cols, size_t c0, size_t c1, size_t ncols,
                                                       size_t row0, size_t nrows, float *dst) {
  // columns [c0, c1) are all flat FLOAT and (c1-c0) % 8 == 0; rows in blocks of 8
  const size_t full = nrows & ~size_t(7);
  for (size_t r = 0; r < full; r += 8)
    for (size_t c = c0; c < c1; c += 8) {
      const float *in[8];
      for (int j = 0; j < 8; j++) in[j] = static_cast<const float *>(cols[c + size_t(j)].data) + row0 + r;
      transpose8x8(in, dst + r * ncols + c, ncols);
    }
  for (size_t r = full; r < nrows; r++)
    for (size_t c = c0; c < c1; c++) dst[r * ncols + c] = static_cast<const float *>(cols[c].data)[row0 + r];
}

}  // namespace

void gather_columns(const infera::InferaColumn *cols, size_t ncols, size_t row0, size_t nrows, float *dst) {
  static const bool have_avx2 = __builtin_cpu_supports("avx2");
  size_t c = 0;
  while (c < ncols) {
    // maximal run of flat FLOAT columns, in multiples of 8 -> AVX2 block transpose
    size_t e = c;
    while (e < ncols && cols[e].type == infera::INFERA_COL_FLOAT && !cols[e].is_constant) e++;
    const size_t run8 = have_avx2 ? ((e - c) & ~size_t(7)) : 0;
    if (run8) gather_float_avx2(cols, c, c + run8, ncols, row0, nrows, dst);
    c += run8;
    if (c < ncols && (c < e || true)) {
      // one leftover / non-FLOAT / constant column, cache-blocked over rows
      const infera::InferaColumn &col = cols[c];
      for (size_t r = 0; r < nrows; r++) dst[r * ncols + c] = cell(col, row0 + r);
      c++;
    }
  }
}

}  // namespace infera_hip
