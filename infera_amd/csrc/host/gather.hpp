// gather.hpp -- columnar DataChunk -> row-major f32 feature matrix (the job of ExtractFeatures,
// infera_extension.cpp:199-227, without the per-cell Value boxing).
#pragma once

#include <cstddef>
#include <cstdint>

#include "../../../include/infera_hip.h"

namespace infera_hip {

// Writes rows [row0, row0+nrows) of the chunk described by `cols` into dst[nrows x ncols] (row-major).
// FLOAT columns go through an 8x8 AVX2 block transpose; DOUBLE / INTEGER / BIGINT (and constant
// vectors) through a cache-blocked scalar loop with static_cast<float> (round-to-nearest-even, the
// reference's casts at infera_extension.cpp:212-214).  Validity is NOT checked here.
void gather_columns(const infera::InferaColumn *cols, size_t ncols, size_t row0, size_t nrows, float *dst);

}  // namespace infera_hip
