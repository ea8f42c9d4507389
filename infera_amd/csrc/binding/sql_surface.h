/*
 * sql_surface.h -- the SQL scalar-function layer of the Infera extension over a MOCK columnar chunk.
 *
 * In production this layer is infera/bindings/infera_extension.cpp inside DuckDB (SURVEY.md section 8a
 * rows 1-8).  DuckDB's headers are not available in the build image (external/duckdb is an empty
 * submodule), so the same functions -- same names, argument checks, NULL behaviour, error strings
 * and output formatting -- are implemented here over a minimal stand-in for DataChunk/Vector
 * (flat or constant typed vectors + validity bitmask, <= 2048 rows per call), so that gather/scatter
 * is testable and benchmarkable without DuckDB.  It calls the engine ONLY through the C ABI of
 * include/infera.h / include/infera_hip.h, exactly like the real binding would.
 *
 * Differences from the reference binding, on purpose (SURVEY.md section 8b "Extensions"):
 *   - ExtractFeatures reads columns through flat pointers + validity masks and hands them to
 *     infera_predict_columns (no per-cell Value boxing; infera_extension.cpp:204-225);
 *   - feature overloads up to INFERA_SQL_MAX_FEATURES (reference: 127, infera_extension.cpp:550);
 *   - infera_predict_array = alias of infera_predict_multi_list;
 *   - PredictFromBlob makes one batched FFI call per chunk when every blob holds one sample.
 */
#ifndef INFERA_SQL_SURFACE_H
#define INFERA_SQL_SURFACE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INFERA_SQL_MAX_FEATURES 1024
#define INFERA_SQL_VECTOR_SIZE 2048 /* DuckDB STANDARD_VECTOR_SIZE */
#define INFERA_SQL_ROW_GROUP 122880 /* DuckDB's default row-group size = 60 vectors */

typedef enum InferaSqlType {
  INFERA_SQL_VARCHAR = 0,
  INFERA_SQL_FLOAT = 1,
  INFERA_SQL_DOUBLE = 2,
  INFERA_SQL_INTEGER = 3,
  INFERA_SQL_BIGINT = 4,
  INFERA_SQL_BLOB = 5,
  INFERA_SQL_BOOLEAN = 6,
  INFERA_SQL_LIST_FLOAT = 7
} InferaSqlType;

/* One argument vector of the chunk (non-owning view, like a DuckDB Vector over its buffer). */
typedef struct InferaSqlVector {
  int32_t type;             /* InferaSqlType */
  int32_t is_constant;      /* CONSTANT_VECTOR: entry 0 applies to every row */
  const void *data;         /* numeric: typed array; VARCHAR/BLOB: const uint8_t *const * (one pointer per row) */
  const uint64_t *lens;     /* VARCHAR/BLOB byte lengths per row */
  const uint64_t *validity; /* bit set = valid; NULL = all valid */
} InferaSqlVector;

/* Result vector (owning).  Exactly one payload is filled according to `type`. */
typedef struct InferaSqlResult {
  int32_t status;         /* 0 ok; -1: `error` holds "Invalid Input Error: ..." as DuckDB would report it */
  char *error;
  int32_t type;           /* InferaSqlType of the result */
  int32_t is_constant;    /* CONSTANT_VECTOR result (entry 0) */
  uint64_t rows;
  float *f32;             /* FLOAT */
  uint8_t *boolean;       /* BOOLEAN */
  char **strings;         /* VARCHAR (rows entries, or 1 if constant) */
  uint64_t *list_offsets; /* LIST<FLOAT>: rows+1 offsets into list_values */
  float *list_values;
  uint64_t *validity;     /* result NULL mask, NULL = all valid */
} InferaSqlResult;

/* Executes SQL scalar function `function` on one chunk.  Returns out->status. */
int32_t infera_sql_call(const char *function, const InferaSqlVector *args, uintptr_t nargs, uintptr_t rows,
                        InferaSqlResult *out);
void infera_sql_free_result(InferaSqlResult *res);
/* JSON list of the registered functions {"name","min_args","max_args","returns","volatile"}; free() it. */
char *infera_sql_list_functions(void);

/* Table-scan driver: `threads` worker threads (DuckDB pipeline workers) pull 2048-row chunks of a
 * synthetic columnar table (ncols FLOAT columns, generator of SURVEY.md 8d) from a shared counter and
 * run `SELECT function(model, c1..cN)` on each through infera_sql_call -- i.e. gather -> C ABI ->
 * H2D -> kernels -> D2H -> result vector, everything the SQL path does per chunk.  Chunk columns are
 * drawn from a per-thread pool of `pool_chunks` pre-generated chunks (table generation is not part of
 * the path).  Returns wall seconds for `rows` rows (<0 on error, message in err); checksum (optional)
 * receives the f64 sum of every output element. */
double infera_sql_bench_scan(const char *function, const char *model, uint64_t rows, uint32_t ncols, int32_t threads,
                             int32_t pool_chunks, uint64_t seed, double *checksum, char *err, uint64_t errlen);

/* The same scan over a MATERIALISED columnar table held in host memory (what bench.py's `end_to_end` block times):
 * row groups of INFERA_SQL_ROW_GROUP rows, one contiguous run per column inside a group -- DuckDB's storage shape, so
 * every chunk column is an 8 KiB run somewhere in a multi-GB table rather than a cache-resident pool.
 *   infera_sql_table_floats   number of floats such a table needs
 *   infera_sql_synth_table    fills it with the generator of SURVEY.md 8d (value of (row, col) independent of layout)
 *   infera_sql_bench_scan_table  runs `reps` complete scans with `threads` workers pulling 2048-row chunks from a
 *                             shared counter; secs[rep] = wall time from the first chunk's gather to the last result
 *                             element consumed.  Table generation is outside every timed region.  0 / -1 (+err). */
uint64_t infera_sql_table_floats(uint64_t rows, uint32_t ncols);
void infera_sql_synth_table(float *table, uint64_t seed, uint64_t rows, uint32_t ncols, int32_t threads);
/* The same with a table of DOUBLE columns (DuckDB's default floating type; the values are the FLOAT table's, widened) */
void infera_sql_synth_table_f64(double *table, uint64_t seed, uint64_t rows, uint32_t ncols, int32_t threads);
int32_t infera_sql_bench_scan_table_typed(const char *function, const char *model, const void *table, int32_t elem_type, uint64_t rows,
                                          uint32_t ncols, int32_t threads, int32_t reps, double *secs, double *checksum, char *err,
                                          uint64_t errlen);
/* The BLOB path's scan (config C5): rows cycle over `nblobs` host-resident blobs of `blob_bytes` each, 2048-row chunks through
 * infera_sql_call("infera_predict_from_blob"); secs[rep] = wall time of scan `rep`.  0 / -1 (+err). */
int32_t infera_sql_bench_blob_scan(const char *model, const uint8_t *blobs, uint64_t nblobs, uint64_t blob_bytes, uint64_t rows,
                                   int32_t threads, int32_t reps, double *secs, double *checksum, char *err, uint64_t errlen);
/* after infera_sql_bench_scan_table: ns spent inside infera_sql_call / inside the worker loops, summed over threads and reps */
void infera_sql_bench_last_times(uint64_t *call_ns, uint64_t *thread_ns);
/* process CPU seconds (user + system over every thread), the system share, and wall seconds of the last bench_scan_table call,
 * summed over its reps: CPU time per chunk is what a cgroup CPU quota meters when 16 CPUs feed 8 GPUs */
void infera_sql_bench_last_cpu(double *cpu_seconds, double *sys_seconds, double *wall_seconds);
int32_t infera_sql_bench_scan_table(const char *function, const char *model, const float *table, uint64_t rows, uint32_t ncols,
                                    int32_t threads, int32_t reps, double *secs, double *checksum, char *err, uint64_t errlen);


/* The host side of the staged scan ALONE: the same workers and chunk order over the same table, each chunk only gathered into the worker's own
 * buffer (infera_gather_columns_colmajor), nothing sent to a GPU.  secs[rep] = wall seconds of scan `rep`; *cpu_seconds = process CPU time over
 * all reps.  rows * ncols * 4 / secs = what the host's memory system delivers to staging buffers at `threads` threads.  0 / -1. */
int32_t infera_sql_bench_gather_only(const float *table, uint64_t rows, uint32_t ncols, int32_t threads, int32_t reps, double *secs, double *cpu_seconds);

/* The same table in DuckDB's SEGMENT shape (round 6): every (row group, column) in ceil(rows_in_group / seg_values) separately allocated
 * blocks of `block_bytes` (256 KiB) whose first `header_bytes` (8) are the block header; seg_values = (block_bytes - header_bytes) / 4.
 * Blocks come from `alloc_fn` (the stub's DBConfig::allocator -- the extension's REGISTERING allocator when INFERA_ZERO_COPY_ALLOCATOR=1) and
 * go back through `free_fn`.  Values are those of infera_sql_synth_table for the same seed.  The scan points each chunk's FLAT vectors into
 * the blocks; the one vector per row group that straddles two segments is assembled in the worker's own buffer (`assembled_chunks`). */
typedef void *(*infera_sql_block_alloc_fn)(void *ctx, uint64_t bytes);
typedef void (*infera_sql_block_free_fn)(void *ctx, void *block, uint64_t bytes);
typedef struct InferaSqlSegmentTable InferaSqlSegmentTable;
InferaSqlSegmentTable *infera_sql_segment_table_create(uint64_t rows, uint32_t ncols, uint64_t seed, int32_t threads, uint64_t block_bytes,
                                                       uint64_t header_bytes, infera_sql_block_alloc_fn alloc_fn, infera_sql_block_free_fn free_fn,
                                                       void *alloc_ctx, uint64_t shuffle_seed /* 0: a row group's segments back to back; else: all blocks in a random order */,
                                                       int32_t alloc_threads /* > 1 (in-order only): that many threads allocate whole (row group, segment) sets concurrently */);
void infera_sql_segment_table_destroy(InferaSqlSegmentTable *t);
uint64_t infera_sql_segment_table_blocks(const InferaSqlSegmentTable *t);
int32_t infera_sql_bench_scan_segments(const char *function, const char *model, const InferaSqlSegmentTable *t, uint64_t rows, int32_t threads,
                                       int32_t reps, double *secs, double *checksum, uint64_t *assembled_chunks, char *err, uint64_t errlen);

#ifdef __cplusplus
}
#endif
#endif
