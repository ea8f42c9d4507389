// mlp_jit.hpp -- load-time specialisation of the fused-MLP device code (mlp_device.inc) with hipRTC for
// chain shapes that have no ahead-of-time instantiation.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace infera_hip::kern {

// Compiles `src` with hipRTC (dlopen'd) for the current device's architecture; returns the code object and the
// lowered symbol of the name expression `expr`.  False + `why` when hipRTC is missing or the compile fails.
bool jit_compile(const char *src, const char *file, const std::string &expr, std::vector<char> &code, std::string &lowered,
                 std::string &why);

// True if a kernel for `sh` is (or was just) compiled; on failure `why` says what went wrong (shape
// constraints, hipRTC missing, compile error) and the caller falls back to layer-by-layer kernels.
bool mlp3_jit_prepare(const Mlp3Shape &sh, std::string *why);
// rows <= 32768 run on the shape's tile kernel when it has one (column-major X only there)
bool mlp3_jit_launch(hipStream_t s, const Mlp3Shape &sh, const float *X, const float *packed, float *Y, int64_t rows,
                     int num_cus, std::string *why, bool x_colmajor = false);
int64_t mlp3_jit_colmajor_max_rows(const Mlp3Shape &sh);  // 0: no column-major kernel
// Chunks of up to this many rows run on the 16-row tile kernel (INFERA_MLP_TILE16_MAX_ROWS; default 4096; 0: never): ONE knob for the
// ahead-of-time configurations (mlp_fused.hip) and the hipRTC ones (mlp_jit.cpp), so the three-way bit-identity test covers both.
int64_t mlp3_tile16_max_rows();
std::string mlp3_jit_kernel_name(const Mlp3Shape &sh);

}  // namespace infera_hip::kern
