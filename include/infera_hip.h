/*
 * infera_hip.h -- ADDITIVE entry points of the MI355X backend (same library, libinfera.so).
 *
 * Nothing here exists in the reference (its roadmap lists "GPU support", "zero-copy" and
 * "BLOB columns in a single FFI call" as not done, /root/reference ROADMAP.md:43-46); the 13
 * reference symbols in infera.h keep their exact signatures.  Everything is plain C: pointers,
 * sizes, status codes (0 ok / -1 error + infera_last_error()).
 */
#ifndef INFERA_HIP_H
#define INFERA_HIP_H

#include "infera.h"

#ifdef __cplusplus
namespace infera {
extern "C" {
#endif

/* ---- device discovery -------------------------------------------------------------------- */

/* Number of GPUs the backend will use (INFERA_DEVICES, default all visible).  0 = none: models
 * still load (metadata, validation) but every predict fails with "ONNX error: HIP backend
 * unavailable: ...".  There is no CPU execution path in this library. */
int32_t infera_hip_device_count(void);
/* HIP ordinal of the i-th selected device, or -1. */
int32_t infera_hip_device_ordinal(int32_t i);
/* JSON: {"devices":[{"ordinal":0,"arch":"gfx950:...","cus":256},..],"reason":".."}.  infera_free. */
char *infera_hip_get_devices(void);
/* JSON description of the lowered plan of a loaded model (steps, fusion decisions, kernel names,
 * flop/row).  infera_free. */
char *infera_hip_get_plan(const char *model_name);

/* ---- device-resident scan (the path bench.py measures) ------------------------------------ */

/* d_in: row-major [rows x cols] f32 in the HBM of HIP device `device`; d_out: caller-allocated
 * [rows x out_cols] f32 on the same device, out_capacity in elements.  Same validation and error
 * strings as infera_predict (engine.rs:111-164).  The kernels are ENQUEUED on the calling thread's
 * stream for that device; call infera_hip_sync before reading d_out.  out_rows/out_cols receive
 * shape_rows_cols of the output (may be NULL).  For models whose input rank is not 2 (images) the
 * buffer holds `rows` samples of cols = prod(input_shape[1:]) elements (the BLOB rule, engine.rs:233-238). */
int32_t infera_hip_predict_device(const char *model_name, int32_t device, const float *d_in, uint64_t rows,
                                  uint64_t cols, float *d_out, uint64_t out_capacity, uint64_t *out_rows,
                                  uint64_t *out_cols);
int32_t infera_hip_sync(int32_t device);

/* Times `iters` back-to-back infera_hip_predict_device passes with HIP events recorded on the
 * stream the kernels are launched on (the calling thread's stream); writes the elapsed
 * milliseconds of all iters to *elapsed_ms.  Used for roofline.achieved in bench.py. */
int32_t infera_hip_time_predict_device(const char *model_name, int32_t device, const float *d_in, uint64_t rows,
                                       uint64_t cols, float *d_out, uint64_t out_capacity, int32_t iters,
                                       float *elapsed_ms);

/* Device-memory helpers so a harness (ctypes, cgo, JNI) needs no second HIP binding. */
void *infera_hip_malloc(int32_t device, uint64_t bytes);
int32_t infera_hip_free(int32_t device, void *ptr);
int32_t infera_hip_memcpy_h2d(int32_t device, void *dst, const void *src, uint64_t bytes);
int32_t infera_hip_memcpy_d2h(int32_t device, void *dst, const void *src, uint64_t bytes);
/* Fills d_dst[rows x cols] with the synthetic table of SURVEY.md 8d / BASELINE.md 4:
 * u = splitmix64(seed ^ (row*cols + col)); x = ((u >> 40) * 2^-24) * 2 - 1, rows starting at row0. */
int32_t infera_hip_synth_fill(int32_t device, float *d_dst, uint64_t seed, uint64_t row0, uint64_t rows,
                              uint64_t cols);

/* ---- host-side fast paths above the reference ABI ------------------------------------------ */

/* infera_predict into a caller-owned buffer (no result allocation/free pair per DataChunk). */
int32_t infera_predict_into(const char *model_name, const float *data, uint64_t rows, uint64_t cols, float *out,
                            uint64_t out_capacity, uint64_t *out_rows, uint64_t *out_cols);

/* Columnar gather: replaces the per-cell Vector::GetValue loop of ExtractFeatures
 * (infera_extension.cpp:199-227).  Each column is a flat typed vector as DuckDB holds it
 * (UnifiedVectorFormat data pointer + validity bitmask, bit set = valid, NULL mask = all valid).
 * Casts follow the reference: DOUBLE/INTEGER/BIGINT -> static_cast<float>.  A NULL cell fails with
 * "Feature values cannot be NULL" (infera_extension.cpp:207-209). */
typedef enum InferaColumnType {
  INFERA_COL_FLOAT = 0,
  INFERA_COL_DOUBLE = 1,
  INFERA_COL_INTEGER = 2,
  INFERA_COL_BIGINT = 3
} InferaColumnType;

typedef struct InferaColumn {
  const void *data;         /* rows elements of the column's type (or 1 element if is_constant) */
  const uint64_t *validity; /* may be NULL */
  int32_t type;             /* InferaColumnType */
  int32_t is_constant;      /* CONSTANT_VECTOR: element 0 applies to every row */
} InferaColumn;

struct InferaInferenceResult infera_predict_columns(const char *model_name, const InferaColumn *columns,
                                                    uintptr_t ncols, uintptr_t rows);

/* The gather step alone (host only, no GPU): rows [row0, row0+nrows) of the chunk -> dst[nrows x ncols]
 * row-major f32.  For bindings that keep calling infera_predict with their own buffer but want the
 * vectorised ExtractFeatures.  Validity masks are checked ("Feature values cannot be NULL"). 0 / -1. */
int32_t infera_gather_columns(const InferaColumn *columns, uintptr_t ncols, uintptr_t row0, uintptr_t nrows, float *dst);
/* The same rows as ONE COLUMN-MAJOR chunk dst[ncols x nrows] -- the layout the host path stages (each column's run copied / converted as
 * it lies: four runs in lockstep, 512 bytes of each in turn) and the fused kernels read.  Host only; what the staged path's gather step does,
 * exposed so that it can be checked without a GPU.  0 / -1. */
int32_t infera_gather_columns_colmajor(const InferaColumn *columns, uintptr_t ncols, uintptr_t row0, uintptr_t nrows, float *dst);

/* One call for a whole chunk of BLOBs (the reference makes one FFI call and one batch-1 run per
 * row, infera_extension.cpp:303-326).  Every blob must hold exactly one sample
 * (prod(input_shape[1:]) f32); NULL entries are not allowed here (the binding filters them).
 * Result: rows = n, cols = per-sample output size. */
struct InferaInferenceResult infera_predict_from_blob_batch(const char *model_name, const uint8_t *const *blobs,
                                                            const uintptr_t *lens, uintptr_t n);

/* The product's own output-shape rule, exposed so it can be pinned on the reference's unit-test table without a GPU
 * (engine.rs:19-29 shape_rows_cols, table at engine.rs:321-328): [] -> (1,1); [n] -> (n,1); [d0,...] ->
 * (d0, max(prod(rest),1)).  infera_predict* report rows/cols of every result through exactly this function. */
void infera_hip_shape_rows_cols(const uint64_t *shape, uintptr_t rank, uint64_t *rows, uint64_t *cols);

/* Measurement helper: GB/s that plain pinned hipMemcpyAsync host->device transfers reach on this box (`threads` threads x
 * `iters` transfers of `bytes`, each on its own stream).  The ceiling bench.py's end_to_end block quotes beside the
 * 64 GB/s raw PCIe Gen5 x16 figure.  < 0 on failure. */
double infera_hip_h2d_probe(int32_t device, uint64_t bytes, int32_t iters, int32_t threads);

/* The policy that deals caller threads over device slots, exposed so it can be tested without 8 GPUs: slots on the
 * thread's own NUMA node first (round-robin among them by `ticket_on_node`), else round-robin over all slots by
 * `ticket_global`.  slot_numa[i] = NUMA node of slot i's GPU (-1 unknown); thread_node < 0 = unknown. */
int32_t infera_hip_choose_slot(const int32_t *slot_numa, uintptr_t nslots, int32_t thread_node, uint64_t ticket_on_node,
                               uint64_t ticket_global);
/* ---- zero-copy host path (round 3; the reference's ROADMAP.md:44 "zero-copy") -------------------------------------------------
 * An application that OWNS long-lived column storage (an in-memory table, Arrow buffers, a DuckDB build with an allocator hook)
 * registers it once: [base, base + bytes) is pinned where it lies and mapped into every selected GPU (hipHostRegister; ~15 us per
 * 480 KB, nothing is copied).  From then on an infera_predict_columns call whose column runs ALL lie inside registered ranges is
 * served without the CPU touching the data: the GPU fetches the runs in place over PCIe -- one 2-D copy when they are FLOAT at one
 * stride inside one registered block (at most three such copies in flight per GPU: the runtime runs them one at a time), else one
 * kernel (FLOAT as they are; DOUBLE / INTEGER / BIGINT converted with the reference's static_cast<float> roundings; constant vectors
 * broadcast) -- into the column-major chunk the model's first kernel reads.  Results are bit-identical to the staged path.  Chunks with any column outside a registered range, and
 * calls longer than one staging pass (> 24 MB of features), take the staged path as before.
 * CONTRACT: a registered range must stay mapped until it is unregistered.  The library never registers memory on its own -- a buffer
 * the caller frees behind a stale registration would fault the GPU.  Ranges must not overlap each other; they MAY share memory pages
 * (neighbours on the heap): the runtime pins whole pages, so ranges whose page spans touch share one registration, which lives until
 * the last of them is unregistered.  Unregistering waits for the zero-copy calls that are reading the pages it unmaps (a chunk's time).  0 / -1 (+ infera_last_error).
 * ONLY after a 0 may the memory be freed or unmapped: -1 with "host memory range still in use" means calls were still reading the range after
 * 5 s (a wedged GPU) -- it is no longer served, but its pages stay pinned and may still be read; keep it mapped. */
int32_t infera_hip_register_host_memory(const void *base, uint64_t bytes);
int32_t infera_hip_unregister_host_memory(const void *base);
/* infera_predict_columns calls served zero-copy so far (tests, bench) */
uint64_t infera_hip_zero_copy_calls(void);

/* The load-aware form the library applies to a caller thread's first call (INFERA_NUMA_SLOTS=1, default): the least-loaded slot
 * on the thread's NUMA node unless it already carries more than one thread above the least-loaded slot overall -- then that one.
 * slot_threads[i] = caller threads currently homed on slot i.  Pure function, exposed for tests. */
int32_t infera_hip_choose_slot_balanced(const int32_t *slot_numa, const int32_t *slot_threads, uintptr_t nslots, int32_t thread_node);

/* sha256(data) as 64 lower-case hex characters: the key under which infera_load_model("http://...") caches a
 * remote model (`<cache_dir>/<sha256(url)>.onnx`, reference http.rs:186-190).  Free with infera_free. */
char *infera_hip_sha256_hex(const char *data, uintptr_t len);

#ifdef __cplusplus
} /* extern "C" */
} /* namespace infera */
#endif

#endif /* INFERA_HIP_H */
