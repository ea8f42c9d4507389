// gather.hpp -- columnar DataChunk -> f32 feature matrix in pinned staging (the job of ExtractFeatures,
// infera_extension.cpp:199-227, without the per-cell Value boxing): column-major (the host path's default: each column's run
// converted or copied as it lies) or row-major (AVX2 transposing gather; infera_gather_columns and the fallback).
#pragma once

#include <cstddef>
#include <cstdint>

#include "../../../include/infera_hip.h"

namespace infera_hip {

// Writes rows [row0, row0+nrows) of the chunk described by `cols` into dst[nrows x ncols] (row-major).
// FLOAT columns go through an 8x8 AVX2 block transpose; DOUBLE / INTEGER / BIGINT (and constant
// vectors) through a cache-blocked scalar loop with static_cast<float> (round-to-nearest-even, the
// reference's casts at infera_extension.cpp:212-214).  Validity is NOT checked here.
// Column-major counterpart: columns [c0, c1) of rows [row0, row0 + nrows) -> dst[c * nrows + r] as f32 (any column type,
// constant vectors filled): what the host path stages when the model's first kernel reads column-major chunks.
void gather_column_major(const infera::InferaColumn *cols, size_t c0, size_t c1, size_t row0, size_t nrows, float *dst);
void gather_columns(const infera::InferaColumn *cols, size_t ncols, size_t row0, size_t nrows, float *dst);

}  // namespace infera_hip
