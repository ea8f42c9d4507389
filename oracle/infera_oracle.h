/*
 * infera_oracle.h -- CPU ORACLE for the infera_predict hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of what the reference computes on the path
 *   infera_load_model -> infera_predict / infera_predict_from_blob
 * (reference: infera/src/engine.rs:19-29, 47-82, 111-164, 199-263; arithmetic = the ONNX
 * operator specification, which the reference delegates to the third-party crate
 * tract-onnx "0.22" (infera/Cargo.toml:21) whose source is NOT in /root/reference).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call
 * this.  The product library (infera_amd/) never does.
 *
 * PARITY PIN STATUS: pinned against every golden value the reference's own tests hold for
 * this path (linear.onnx (1,2,3)->1.75, (0,0,0)->0.25; multi_output.onnx identity; the
 * blob/shape error strings).  For Gemm/Relu/Sigmoid/Softmax/Conv/BatchNorm/pooling and for
 * any chunk with more than one row the reference has no test and Tract cannot be built
 * here (no cargo/rustc, crate not vendored): PARITY UNPINNED for those operators -- the
 * oracle follows the ONNX operator spec.
 */
#ifndef INFERA_ORACLE_H
#define INFERA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcModel OrcModel;

/* Mirrors ffi_utils.rs:10-22 (InferaInferenceResult) minus the status field. */
typedef struct OrcResult {
  float *data;
  size_t len;
  size_t rows;
  size_t cols;
} OrcResult;

/* engine.rs:47-82 load_model_impl: parse, validate, extract input/output shape (-1 = symbolic). */
OrcModel *orc_load(const char *path, char *err, size_t errlen);
void orc_free_model(OrcModel *m);
int orc_input_rank(const OrcModel *m);
int orc_output_rank(const OrcModel *m);
const int64_t *orc_input_shape(const OrcModel *m);
const int64_t *orc_output_shape(const OrcModel *m);

/* engine.rs:19-29 shape_rows_cols */
void orc_shape_rows_cols(const size_t *shape, int rank, size_t *rows, size_t *cols);

/* engine.rs:111-164 run_inference_impl (model lookup excluded).  0 ok / -1 error (err holds the
 * InferaError Display text, error.rs:13-61). */
int orc_predict(const OrcModel *m, const float *data, size_t rows, size_t cols, OrcResult *out,
                char *err, size_t errlen);
/* engine.rs:199-263 run_inference_blob_impl */
int orc_predict_blob(const OrcModel *m, const uint8_t *blob, size_t len, OrcResult *out,
                     char *err, size_t errlen);
void orc_free_result(OrcResult *r);

/* Synthetic table generator shared by oracle, HIP fill kernel and numpy:
 *   u = splitmix64(seed ^ (row*F + col)); x = ((u >> 40) * 2^-24) * 2 - 1   (exact in f32) */
uint64_t orc_splitmix64(uint64_t x);
float orc_synth_value(uint64_t seed, uint64_t row, uint64_t col, uint64_t ncols);
void orc_synth_fill_rowmajor(float *dst, uint64_t seed, uint64_t row0, uint64_t rows,
                             uint64_t ncols);

/* CPU baseline ("port"): T worker threads pull 2048-row chunks of a synthetic columnar table,
 * gather to row-major f32 (boxed=1: per-cell tagged-value path in the cost class of
 * Vector::GetValue, infera_extension.cpp:204-225; boxed=0: contiguous column reads), run the whole
 * graph single-threaded per chunk (engine.rs:139-154) and copy the result out
 * (infera_extension.cpp:280-284).  Returns seconds of wall time for `rows` rows, <0 on error.
 * checksum (optional) receives the f64 sum of all outputs. */
double orc_bench_scan(const OrcModel *m, uint64_t rows, uint64_t ncols, uint64_t seed, int threads,
                      int chunk_rows, int boxed, double *checksum);

/* The same scan over a table MATERIALISED by the caller in host memory (row groups of 122,880 rows, one
 * contiguous run per column inside a group -- infera_amd/csrc/binding/sql_surface.h uses the same layout), so
 * that table generation is outside the timed region and the CPU baseline and the GPU path read
 * the very same bytes.  chunk_rows must divide 122,880.
 * boxed: 1 = the reference's per-cell boxed gather + plain GEMM loop ("reference-shaped"); 0 = plain strided gather, same GEMM;
 * 2 = "best CPU" (BASELINE.md 3b): tiled transposing gather + register-blocked AVX-512 / AVX2 micro-kernel GEMM -- bit-identical
 * results (every element stays one k-ordered fmaf chain), the instruction schedule of a packed SIMD matmul such as Tract's. */
double orc_bench_scan_table(const OrcModel *m, const float *table, uint64_t rows, uint64_t ncols,
                            int threads, int chunk_rows, int boxed, double *checksum);
/* test hook: later orc_predict calls ON THE CALLING THREAD use the blocked GEMM (1) or the plain loop (0, default) */
void orc_set_blocked_gemm(int on);

#ifdef __cplusplus
}
#endif
#endif
