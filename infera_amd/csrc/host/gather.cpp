#include "gather.hpp"

#include "common.hpp"

#include <immintrin.h>

#include <cstring>
#include <vector>

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

// 8 consecutive rows of one column as f32, whatever its type (casts as the reference: static_cast<float>,
// infera_extension.cpp:211-222).  DOUBLE and INTEGER convert in registers; BIGINT has no AVX2 conversion.
__attribute__((target("avx2"))) inline const float *rows8_as_float(const infera::InferaColumn &c, size_t row, float *tmp) {
  if (c.is_constant) {
    _mm256_storeu_ps(tmp, _mm256_set1_ps(cell(c, 0)));
    return tmp;
  }
  switch (c.type) {
    case infera::INFERA_COL_FLOAT: return static_cast<const float *>(c.data) + row;
    case infera::INFERA_COL_DOUBLE: {
      const double *p = static_cast<const double *>(c.data) + row;
      _mm_storeu_ps(tmp, _mm256_cvtpd_ps(_mm256_loadu_pd(p)));
      _mm_storeu_ps(tmp + 4, _mm256_cvtpd_ps(_mm256_loadu_pd(p + 4)));
      return tmp;
    }
    case infera::INFERA_COL_INTEGER:
      _mm256_storeu_ps(tmp, _mm256_cvtepi32_ps(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(static_cast<const int32_t *>(c.data) + row))));
      return tmp;
    default: {
      const int64_t *p = static_cast<const int64_t *>(c.data) + row;
      for (int i = 0; i < 8; i++) tmp[i] = static_cast<float>(p[i]);
      return tmp;
    }
  }
}

// One 8-column block whose columns are all flat and of ONE type: no per-cell dispatch.
template <int TYPE>
__attribute__((target("avx2"))) inline void block_uniform(const infera::InferaColumn *cols, size_t row, float *dst, size_t ld) {
  alignas(32) float tmp[8][8];
  const float *in[8];
  for (int j = 0; j < 8; j++) {
    if constexpr (TYPE == infera::INFERA_COL_FLOAT) {
      in[j] = static_cast<const float *>(cols[j].data) + row;
    } else if constexpr (TYPE == infera::INFERA_COL_DOUBLE) {
      const double *p = static_cast<const double *>(cols[j].data) + row;
      _mm_store_ps(tmp[j], _mm256_cvtpd_ps(_mm256_loadu_pd(p)));
      _mm_store_ps(tmp[j] + 4, _mm256_cvtpd_ps(_mm256_loadu_pd(p + 4)));
      in[j] = tmp[j];
    } else if constexpr (TYPE == infera::INFERA_COL_INTEGER) {
      const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(static_cast<const int32_t *>(cols[j].data) + row));
      _mm256_store_ps(tmp[j], _mm256_cvtepi32_ps(v));
      in[j] = tmp[j];
    } else {
      const int64_t *p = static_cast<const int64_t *>(cols[j].data) + row;
      for (int i = 0; i < 8; i++) tmp[j][i] = static_cast<float>(p[i]);
      in[j] = tmp[j];
    }
  }
  transpose8x8(in, dst, ld);
}

// Columns [0, blocked), blocked % 8 == 0: 8x8 blocks, each written as eight 32-byte row pieces (the scalar path
// writes one float per 4*ncols-byte stride).  kind[b]: the block's uniform type, or -1 = mixed / constant vectors.
__attribute__((target("avx2"))) void gather_blocks_avx2(const infera::InferaColumn *cols, size_t blocked, size_t ncols, size_t row0,
                                                        size_t nrows, float *dst) {
  const size_t nb = blocked / 8, full = nrows & ~size_t(7);
  std::vector<int8_t> kind(nb);
  for (size_t b = 0; b < nb; b++) {
    int k = cols[8 * b].type;
    for (int j = 0; j < 8; j++)
      if (cols[8 * b + size_t(j)].is_constant || cols[8 * b + size_t(j)].type != k) k = -1;
    kind[b] = int8_t(k);
  }
  alignas(32) float tmp[8][8];
  for (size_t r = 0; r < full; r += 8)
    for (size_t b = 0; b < nb; b++) {
      const infera::InferaColumn *cb = cols + 8 * b;
      float *d = dst + r * ncols + 8 * b;
      switch (kind[b]) {
        case infera::INFERA_COL_FLOAT: block_uniform<infera::INFERA_COL_FLOAT>(cb, row0 + r, d, ncols); break;
        case infera::INFERA_COL_DOUBLE: block_uniform<infera::INFERA_COL_DOUBLE>(cb, row0 + r, d, ncols); break;
        case infera::INFERA_COL_INTEGER: block_uniform<infera::INFERA_COL_INTEGER>(cb, row0 + r, d, ncols); break;
        case infera::INFERA_COL_BIGINT: block_uniform<infera::INFERA_COL_BIGINT>(cb, row0 + r, d, ncols); break;
        default: {
          const float *in[8];
          for (int j = 0; j < 8; j++) in[j] = rows8_as_float(cb[j], row0 + r, tmp[j]);
          transpose8x8(in, d, ncols);
        }
      }
    }
  for (size_t r = full; r < nrows; r++)
    for (size_t c = 0; c < blocked; c++) dst[r * ncols + c] = cell(cols[c], row0 + r);
}

}  // namespace

namespace {
__attribute__((target("avx2"))) void convert_f64_avx2(const double *src, float *dst, size_t n) {
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    _mm_storeu_ps(dst + i, _mm256_cvtpd_ps(_mm256_loadu_pd(src + i)));
    _mm_storeu_ps(dst + i + 4, _mm256_cvtpd_ps(_mm256_loadu_pd(src + i + 4)));
  }
  for (; i < n; i++) dst[i] = static_cast<float>(src[i]);
}
__attribute__((target("avx2"))) void convert_i32_avx2(const int32_t *src, float *dst, size_t n) {
  size_t i = 0;
  for (; i + 8 <= n; i += 8) _mm256_storeu_ps(dst + i, _mm256_cvtepi32_ps(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i))));
  for (; i < n; i++) dst[i] = static_cast<float>(src[i]);
}
}  // namespace

namespace {
// Several column runs are copied in lockstep, a block of each in turn -- independent sequential streams keep more cache-line fills in
// flight than one (the copy of one 8 KiB run is latency-bound at the start of every run: a new page, a new DRAM row, a hardware
// prefetcher that has to find the stream again).  FOUR streams of 512 bytes: C2's gather 52-58 -> 42-44 us per chunk at 4..24 callers;
// two streams gain little, eight and sixteen LOSE (the runs lie 8 KiB apart in staging: their lines alias in the L1).  Measured and
// dropped (round 3, profiles/r03_gather_interleave_sweep.txt, r03_host_cpu_ab_wait_gather.txt, r03_gather_probe.txt): one run at a
// time (memcpy), non-temporal stores with and without a prefetch of the next run (a reused staging buffer under regular stores stays
// cache-resident; NT stores stop at ~140 GB/s of DRAM writes at 16 threads), interleaved + non-temporal.
constexpr int kIlStreams = 4;
constexpr size_t kIlFloats = 128;  // 512 bytes of a FLOAT run (and of a DOUBLE run: 64 elements) per turn; a multiple of 16
__attribute__((target("avx2"))) void copy_interleaved_f32(float *const *d, const float *const *s, int ns, size_t n, size_t blk) {
  size_t i = 0;
  for (; i + blk <= n; i += blk)
    for (int k = 0; k < ns; k++) {
      const float *sp = s[k] + i;
      float *dp = d[k] + i;
      for (size_t v = 0; v < blk; v += 8) _mm256_storeu_ps(dp + v, _mm256_loadu_ps(sp + v));
    }
  for (int k = 0; k < ns; k++)
    if (i < n) std::memcpy(d[k] + i, s[k] + i, (n - i) * sizeof(float));
}
// the same for DOUBLE runs (DuckDB's default floating type): vcvtpd2ps = static_cast<float>, blk source elements per turn (blk % 8 == 0)
__attribute__((target("avx2"))) void convert_interleaved_f64(float *const *d, const double *const *s, int ns, size_t n, size_t blk) {
  size_t i = 0;
  for (; i + blk <= n; i += blk)
    for (int k = 0; k < ns; k++) {
      const double *sp = s[k] + i;
      float *dp = d[k] + i;
      for (size_t v = 0; v < blk; v += 8) {
        _mm_storeu_ps(dp + v, _mm256_cvtpd_ps(_mm256_loadu_pd(sp + v)));
        _mm_storeu_ps(dp + v + 4, _mm256_cvtpd_ps(_mm256_loadu_pd(sp + v + 4)));
      }
    }
  for (int k = 0; k < ns; k++)
    for (size_t r = i; r < n; r++) d[k][r] = static_cast<float>(s[k][r]);
}
}  // namespace

// Column-major staging: rows [row0, row0 + nrows) of column c -> dst[c * nrows ...] as f32, the column's own run converted in
// place of the plain memcpy a FLOAT column gets (static_cast<float> per the reference, infera_extension.cpp:211-222: RNE for
// DOUBLE -- what vcvtpd2ps does).  DuckDB's default floating type is DOUBLE, so this is the common case of a real table: no
// transposing gather on the CPU, the GPU kernel reads the chunk column-major.
void gather_column_major(const infera::InferaColumn *cols, size_t c0, size_t c1, size_t row0, size_t nrows, float *dst) {
  static const bool have_avx2 = __builtin_cpu_supports("avx2");
  for (size_t c = c0; c < c1; c++) {
    if (have_avx2) {  // several plain FLOAT (or DOUBLE) runs in lockstep
      int ns = 0;
      float *dn[kIlStreams];
      const float *sn[kIlStreams];
      while (ns < kIlStreams && c + size_t(ns) < c1 && !cols[c + size_t(ns)].is_constant && cols[c + size_t(ns)].type == infera::INFERA_COL_FLOAT) {
        dn[ns] = dst + (c + size_t(ns)) * nrows;
        sn[ns] = static_cast<const float *>(cols[c + size_t(ns)].data) + row0;
        ns++;
      }
      if (ns >= 2) {
        copy_interleaved_f32(dn, sn, ns, nrows, kIlFloats);
        c += size_t(ns) - 1;
        continue;
      }
      const double *sd[kIlStreams];
      ns = 0;
      while (ns < kIlStreams && c + size_t(ns) < c1 && !cols[c + size_t(ns)].is_constant && cols[c + size_t(ns)].type == infera::INFERA_COL_DOUBLE) {
        dn[ns] = dst + (c + size_t(ns)) * nrows;
        sd[ns] = static_cast<const double *>(cols[c + size_t(ns)].data) + row0;
        ns++;
      }
      if (ns >= 2) {
        convert_interleaved_f64(dn, sd, ns, nrows, kIlFloats / 2);  // (the same bytes of source per turn)
        c += size_t(ns) - 1;
        continue;
      }
    }
    const infera::InferaColumn &col = cols[c];
    float *d = dst + c * nrows;
    if (col.is_constant) {
      const float v = cell(col, 0);
      for (size_t r = 0; r < nrows; r++) d[r] = v;
      continue;
    }
    switch (col.type) {
      case infera::INFERA_COL_FLOAT: std::memcpy(d, static_cast<const float *>(col.data) + row0, nrows * sizeof(float)); break;
      case infera::INFERA_COL_DOUBLE:
        if (have_avx2) convert_f64_avx2(static_cast<const double *>(col.data) + row0, d, nrows);
        else for (size_t r = 0; r < nrows; r++) d[r] = static_cast<float>(static_cast<const double *>(col.data)[row0 + r]);
        break;
      case infera::INFERA_COL_INTEGER:
        if (have_avx2) convert_i32_avx2(static_cast<const int32_t *>(col.data) + row0, d, nrows);
        else for (size_t r = 0; r < nrows; r++) d[r] = static_cast<float>(static_cast<const int32_t *>(col.data)[row0 + r]);
        break;
      default:
        for (size_t r = 0; r < nrows; r++) d[r] = static_cast<float>(static_cast<const int64_t *>(col.data)[row0 + r]);
    }
  }
}

void gather_columns(const infera::InferaColumn *cols, size_t ncols, size_t row0, size_t nrows, float *dst) {
  static const bool have_avx2 = __builtin_cpu_supports("avx2");
  const size_t blocked = have_avx2 ? (ncols & ~size_t(7)) : 0;
  if (blocked) gather_blocks_avx2(cols, blocked, ncols, row0, nrows, dst);
  for (size_t c = blocked; c < ncols; c++) {  // < 8 leftover columns (or no AVX2)
    const infera::InferaColumn &col = cols[c];
    for (size_t r = 0; r < nrows; r++) dst[r * ncols + c] = cell(col, row0 + r);
  }
}

}  // namespace infera_hip
