/*
 * infera.h -- C ABI of the MI355X-native Infera backend (libinfera.so).
 *
 * This header is the DROP-IN BOUNDARY.  It declares exactly the 13 functions + 1 struct that the
 * reference's Rust crate exports through cbindgen and that its DuckDB binding links against
 * (reference: infera/bindings/include/rust.h:28-49 struct, :77-316 prototypes; export list
 * infera/cbindgen.toml:35-50).  Signatures, struct layout (40 bytes on LP64, passed and returned
 * BY VALUE), status convention (0 / -1), ownership rules and error strings are identical, so
 * infera/bindings/infera_extension.cpp compiles and links against this library unchanged
 * (include this file where it includes "rust.h").  What differs is what happens underneath:
 * infera_load_model lowers the ONNX graph to hand-written gfx950 HIP kernels and uploads the
 * weights to HBM; infera_predict stages rows through pinned memory to the GPU.
 *
 * Additive, MI355X-specific entry points (device-resident scans, columnar gather, batched
 * blobs, profiling hooks) are declared separately in infera_hip.h.
 */
#ifndef INFERA_H
#define INFERA_H

#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
namespace infera {
#endif

/* replaces rust.h:28-49 / ffi_utils.rs:10-22.  On failure: data=NULL, len=rows=cols=0, status=-1
 * (ffi_utils.rs:28-36). */
typedef struct InferaInferenceResult {
  float *data;    /* callee-allocated; release ONLY via infera_free_result */
  uintptr_t len;  /* number of f32 elements in data */
  uintptr_t rows; /* shape_rows_cols(output shape).0  (engine.rs:19-29) */
  uintptr_t cols; /* shape_rows_cols(output shape).1 */
  int32_t status; /* 0 ok, -1 error (text via infera_last_error on the same thread) */
} InferaInferenceResult;

#ifdef __cplusplus
extern "C" {
#endif

/* replaces rust.h:77-78 (lib.rs:38-64).  Parses `path`, lowers the graph to HIP kernels, uploads
 * weights to every selected GPU, registers under `name` (same name silently replaces,
 * engine.rs:74-80).  Paths starting with "http" are fetched into the on-disk model cache first, as in
 * the reference (lib.rs:47-58 -> http.rs:179-335: sha256(url) cache key, ETag revalidation, LRU eviction
 * under INFERA_CACHE_SIZE_LIMIT, retries); failures read "HTTP request failed: ...". 0 / -1.
 * Additive: "<path>#<output>" registers the model with another graph output than the first as the one it serves
 * (output name or decimal index; e.g. "clf.onnx#probabilities" beside "clf.onnx" = the label) -- the reference always
 * serves output 0 (engine.rs:146-149).  A path that exists as written is taken as written. */
int32_t infera_load_model(const char *name, const char *path);

/* replaces rust.h:97 (lib.rs:81-102).  -1 + "Model not found: <name>" if absent. */
int32_t infera_unload_model(const char *name);

/* replaces rust.h:125-128 (lib.rs:127-149 -> engine.rs:111-164).  `data` is row-major
 * [rows x cols] f32 in HOST memory, borrowed for the call only. */
struct InferaInferenceResult infera_predict(const char *model_name, const float *data, uintptr_t rows,
                                            uintptr_t cols);

/* replaces rust.h:156-158 (lib.rs:174-195 -> engine.rs:199-263).  Native-endian f32 bytes. */
struct InferaInferenceResult infera_predict_from_blob(const char *model_name, const uint8_t *blob_data,
                                                      uintptr_t blob_len);

/* replaces rust.h:180 (lib.rs:215-233).  JSON {"input_shape":[..],"loaded":true,"name":"..",
 * "output_shape":[..]} or {"error":".."}.  Free with infera_free. */
char *infera_get_model_info(const char *model_name);

/* replaces rust.h:194 (lib.rs:245-260).  JSON array of names. */
char *infera_get_loaded_models(void);

/* replaces rust.h:211 (lib.rs:275-285).  {"model_cache_dir":..,"onnx_backend":"hip-gfx950","version":..} */
char *infera_get_version(void);

/* replaces rust.h:227 (lib.rs:299-308). */
int32_t infera_clear_cache(void);

/* replaces rust.h:247 (lib.rs:326-366). */
char *infera_get_cache_info(void);

/* replaces rust.h:271 (lib.rs:388-425).  {"loaded":[..],"errors":[{"file":..,"error":..}]} */
char *infera_set_autoload_dir(const char *path);

/* replaces rust.h:285 (error.rs:96-102).  Thread-local, borrowed, NULL if this thread never failed;
 * not cleared by later successes. */
const char *infera_last_error(void);

/* replaces rust.h:299 (ffi_utils.rs:49-54).  NULL is a no-op. */
void infera_free(char *ptr);

/* replaces rust.h:316 (ffi_utils.rs:69-77).  NULL data is a no-op. */
void infera_free_result(struct InferaInferenceResult res);

#ifdef __cplusplus
} /* extern "C" */
} /* namespace infera */
#endif

#endif /* INFERA_H */
