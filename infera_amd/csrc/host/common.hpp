// common.hpp -- error type, thread-local last-error slot, env configuration, JSON helpers.
//
// Mirrors (behaviour, not code) the reference's error plumbing and config singleton:
//   error.rs:9-62   InferaError variants and their Display strings (asserted verbatim by the
//                   reference's SQL tests)
//   error.rs:70-102 thread-local LAST_ERROR / infera_last_error
//   config.rs:101-176 INFERA_* environment variables read once
#pragma once

#include <cstdint>
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace infera_hip {

enum class ErrKind {
  ModelNotFound,
  InvalidInputShape,
  Onnx,
  Memory,
  Utf8,
  NullPointer,
  Io,
  Json,
  FeatureNotEnabled,
  HttpRequest,
  CacheDir,
  InvalidBlobSize,
  BlobShapeMismatch,
};

// An error carrying the exact text the reference's `thiserror` Display would print.
class InferaError : public std::runtime_error {
 public:
  InferaError(ErrKind k, const std::string &text) : std::runtime_error(text), kind(k) {}
  ErrKind kind;

  static InferaError model_not_found(const std::string &name) {
    return {ErrKind::ModelNotFound, "Model not found: " + name};  // error.rs:13-14
  }
  static InferaError invalid_input_shape(const std::string &expected, const std::string &actual) {
    return {ErrKind::InvalidInputShape, "Invalid input shape: expected " + expected + ", got " + actual};  // :16-22
  }
  static InferaError onnx(const std::string &msg) { return {ErrKind::Onnx, "ONNX error: " + msg}; }  // :24-25
  static InferaError memory() { return {ErrKind::Memory, "Memory allocation error"}; }                // :27-28
  static InferaError utf8() { return {ErrKind::Utf8, "Invalid UTF-8 string"}; }                       // :30-31
  static InferaError null_pointer() { return {ErrKind::NullPointer, "Null pointer passed"}; }         // :33-34
  static InferaError io(const std::string &msg) { return {ErrKind::Io, "IO error: " + msg}; }         // :36-37
  static InferaError json(const std::string &msg) { return {ErrKind::Json, "JSON serialization error: " + msg}; }
  static InferaError feature_not_enabled(const std::string &msg) {
    return {ErrKind::FeatureNotEnabled, "Feature not enabled: " + msg};  // :42-43
  }
  static InferaError http(const std::string &msg) { return {ErrKind::HttpRequest, "HTTP request failed: " + msg}; }
  static InferaError cache_dir(const std::string &msg) {
    return {ErrKind::CacheDir, "Failed to create cache directory: " + msg};
  }
  static InferaError invalid_blob_size() {
    return {ErrKind::InvalidBlobSize, "Invalid BLOB size: length must be a multiple of 4"};  // :51-52
  }
  static InferaError blob_shape_mismatch(size_t expected, size_t actual) {  // :54-61
    return {ErrKind::BlobShapeMismatch, "BLOB data does not match model's expected input shape. Expected " +
                                            std::to_string(expected) + " elements, but BLOB contained " +
                                            std::to_string(actual) + "."};
  }
};

// error.rs:78-84 / :96-102.  The slot is per thread, survives later successes, and the returned
// pointer stays valid until the next error on the same thread.
void set_last_error(const std::string &text);
const char *last_error_cstr();

// Valid UTF-8 check (CStr::to_str in lib.rs:44-45, :136).
bool is_valid_utf8(const char *s);

// ---------------------------------------------------------------------------------------------
// Configuration (config.rs:101-176 pattern: env vars read once, invalid values fall back)
// ---------------------------------------------------------------------------------------------
struct Config {
  std::string cache_dir;              // INFERA_CACHE_DIR        (default $TMPDIR/infera_cache)
  uint64_t cache_size_limit;          // INFERA_CACHE_SIZE_LIMIT (default 1 GiB)
  int log_level;                      // INFERA_LOG_LEVEL        ERROR=0 WARN=1 INFO=2 DEBUG=3 (default WARN)
  uint64_t http_timeout_secs;         // INFERA_HTTP_TIMEOUT        (default 30; config.rs:138-144)
  uint32_t http_retry_attempts;       // INFERA_HTTP_RETRY_ATTEMPTS (default 3)
  uint64_t http_retry_delay_ms;       // INFERA_HTTP_RETRY_DELAY    (default 1000, multiplied by the attempt number)
  // MI355X backend knobs (new; same style)
  std::vector<int> devices;           // INFERA_DEVICES="0,1,.."  (default: all visible)
  int host_contexts;                  // INFERA_HOST_CONTEXTS=n   staging contexts (stream + pinned / device buffers + scratch) per GPU that host-ABI calls lease
  int max_inflight;                   // INFERA_MAX_INFLIGHT=n    host-ABI calls per GPU between their first H2D and their sync (0 = no limit)
  bool use_hipgraph;                  // INFERA_HIPGRAPH=0|1      replay a per-(model,rows) hipGraph {H2D,kernels,D2H} per
                                      //   host-path chunk.  Default 0: measured SLOWER than three direct stream
                                      //   enqueues on MI355X/ROCm 7.2 (60 vs 82 M rows/s at 16 threads)
  int host_wait;                      // INFERA_HOST_WAIT=poll|pollq|block|spin  how a host-ABI call waits for its chunk.  poll (2, default) = nap
                                      //   for most of the expected wait (clock_nanosleep), then hipEventQuery between short naps: the caller's
                                      //   core is free for other workers' gathers -- what a CPU-quota'd container or a busy DuckDB pipeline
                                      //   needs; pollq (3) = the same on hipStreamQuery (no event record); block (0) = hipEventSynchronize on a
                                      //   blocking-sync event, which on ROCm 7.2 BURNS the core for the whole wait (measured: 277 us of CPU per
                                      //   chunk at 16 threads against 85 with poll, same rows/s; profiles/r03_host_cpu_ab_wait_gather.txt);
                                      //   spin (1) = hipStreamSynchronize
  double host_poll_first, host_poll_next;  // INFERA_HOST_POLL_FIRST / _NEXT (default 0.75 / 0.1): poll mode naps first for this share of the
                                      //   context's recent wait, then this share between event queries
  bool host_ctx_affinity;             // INFERA_HOST_CTX_AFFINITY=0|1 (default 1)  a caller thread re-leases the staging context it used last when free:
                                      //   its pinned staging lines are still in that core's caches (CPU per chunk 88.9 -> 76.9 us at 16 callers)
  int host_gather;                    // INFERA_HOST_GATHER=il|memcpy|nt|ntpf|ilnt  how FLOAT column runs are copied into pinned staging.  il (3, default
                                      //   since round 3): FOUR runs in lockstep, 512 bytes of each in turn -- four sequential streams keep more
                                      //   line fills in flight than one 8 KiB run after the other (gather 52-58 -> 42-44 us per chunk at 4-24
                                      //   callers; 2 streams gain little, 8 and 16 lose: the runs are 8 KiB apart in staging and alias in L1;
                                      //   INFERA_GATHER_IL_STREAMS / _BYTES for A/B).  memcpy (0) one run at a time; nt (1) non-temporal 64-byte
                                      //   stores, ntpf (2) + prefetch of the next run, ilnt (4) interleaved + non-temporal: all measured worse
                                      //   at scale (profiles/r03_gather_interleave_sweep.txt, r03_host_cpu_ab_wait_gather.txt)
  int max_inflight_total;             // INFERA_MAX_INFLIGHT_TOTAL=n  host-ABI calls the PROCESS admits between first H2D and sync over all
                                      //   GPUs (0 = no process-wide limit; the per-GPU limit is INFERA_MAX_INFLIGHT)
  int probe_elide_h2d;                // INFERA_HOST_PROBE_ELIDE_H2D=1|2  MEASUREMENT ONLY (bench.py --elide-h2d): host-path H2D copies move a
                                      //   4 KiB token instead of the chunk, so the gather / lease / gate / submit machinery can be timed with
                                      //   the link taken out; 2 = the kernels also run on one 32-row tile only (8 slots sharing ONE
                                      //   GPU are otherwise bound by that GPU's kernel dispatch rate).  Results are meaningless in this mode
  bool host_zero_copy;                // INFERA_HOST_ZERO_COPY=0|1 (default 1)  infera_predict_columns chunks whose column runs all lie in host memory
                                      //   registered with infera_hip_register_host_memory are read in place by the GPU (no CPU gather, no H2D copy)
  bool numa_slots;                    // INFERA_NUMA_SLOTS=0|1 (default 1)  caller threads prefer the device slots on their own NUMA node (bounded by load)
  int host_split;                     // INFERA_HOST_SPLIT=0|1|n  one-DataChunk calls go through as sub-passes on the call's stream, the gather
                                      //   of sub-pass i+1 overlapping H2D + kernels of sub-pass i: 0 never (default), 1 two halves when the GPU
                                      //   is quiet (at most INFERA_HOST_SPLIT_QUIET calls in flight), n >= 2 always n sub-passes.  Measured a LOSS
                                      //   at every thread count (C2, 1 / 4 / 8 callers: 20.3 / 61.2 / 92.9 M rows/s whole, 18.5 / 51.4 / 81.2 in
                                      //   halves; profiles/r03_host_cpu_ab_split_pollq.txt): a chunk's 45-50 us in flight are fixed latencies
                                      //   (copy-engine start, dispatch, completion), not its 20 us of transfer -- halves pay them twice
  int host_split_quiet;               // INFERA_HOST_SPLIT_QUIET=n (default 4)
  bool host_direct_out;               // INFERA_HOST_DIRECT_OUT=0|1  the last kernel of a write-once plan stores its results
                                      //   straight into the pinned result buffer (no D2H copy enqueue per chunk)
  bool host_colmajor_typed;           // INFERA_HOST_COLMAJOR_TYPED=0|1  DOUBLE / INTEGER / BIGINT / constant columns are staged column-major too
                                      //   (converted run by run) instead of through the AVX2 transposing gather.  Default 1
  long long host_direct_in_bytes;     // INFERA_HOST_DIRECT_IN=<bytes>  chunks up to this size are read from pinned memory by the first kernel itself
                                      //   (default 131072; 0 = always H2D).  On a quiet GPU the effective limit is higher: x2 with at most four
                                      //   host-ABI calls in flight, x4 (512 KB) with at most two -- INFERA_HOST_DIRECT_IN_QUIET=0 switches that off
  bool host_direct_in_quiet;          // INFERA_HOST_DIRECT_IN_QUIET=0|1 (default 1)  the x2 / x4 multiplier above
  bool mlp3_tile;                     // INFERA_MLP3_TILE=0|1 (default 1)  launches of up to 32,768 rows of a fused MLP take the one-workgroup-per-tile
                                      //   kernel (ahead-of-time and load-time specialised chains alike; read by mlp_fused.hip and mlp_jit.cpp)
  bool host_fused_transpose;          // INFERA_HOST_FUSED_TRANSPOSE=0|1  the fused MLP reads column-major chunks itself (no transpose kernel)
  bool precision_bf16x3;              // INFERA_PRECISION=fp32|bf16x3  bf16x3 = OPTIONAL fast mode for the fused MLP (three bf16 MFMAs per
                                      //   product, ~2^-16 relative error per product): NOT the parity path, never the default
  bool fused_mlp;                     // INFERA_FUSED_MLP=0|1     whole-chain fused kernel when the plan allows
  uint64_t max_rows_per_pass;         // INFERA_MAX_ROWS_PER_PASS scratch bound for unfused plans
  bool batch_split;                   // INFERA_BATCH_SPLIT=0|1   a model with a FIXED leading dim B accepts any multiple
                                      //   of B rows (every supported operator is row-independent).  Default 0 =
                                      //   the reference's behaviour: rows != B is an error (test/models/README.md:5)
  static const Config &get();
};

// Knobs that are read EVERY TIME a model is scheduled (infera_load_model), not once per process, so that one process can load the
// same graph both ways (the bit-identity tests do): each defaults to on.
struct ScheduleKnobs {
  bool stem_pool;   // INFERA_STEM_POOL=0|1   stem convolution + MaxPool 3x3/2 in one kernel
  bool chain_xcm;   // INFERA_CHAIN_XCM=0|1   the fused small-MLP chain kernel reads column-major host chunks itself (0: transpose launch first)
  bool dense_xcm;   // INFERA_DENSE_XCM=0|1   the same for the as-it-lies streaming kernels of single narrow layers
  bool conv_bf16x6; // DEFAULT (INFERA_PRECISION unset / bf16x6): the tiled convolutions on the bf16 matrix cores, every fp32 operand cut EXACTLY into
                    //   three bf16 parts, six partial products per product, fp32 accumulate (conv_split.hip) -- no scales, no precondition on the data.
                    //   INFERA_PRECISION=fp32: the exact-fp32 matrix instruction instead (conv.hip's tiled / weight-stationary kernels)
  bool conv_f16x3;  // INFERA_PRECISION=f16x3  the tiled convolutions on the fp16 matrix cores, operands split hi + lo (conv_split.hip)
  static ScheduleKnobs read();
};
// Read at first use inside the kernel launchers, A/B experiments only (no effect on results; defaults are the shipped paths):
//   INFERA_CONV_WS (0 tiled kernel only | 1 default | 2 force the weight-stationary kernel; the split-fp16 forms of conv_split.hip follow it too),
//   INFERA_STEM_SPLIT (0: the exact-fp32 stem kernels under a default (bf16x6) or f16x3 plan -- tests / A/B only, the forms are not bit-compatible),
//   INFERA_SPLIT_PROBE (PROBES builds: timing probes of the split convolution, wrong results), INFERA_DENSE16W, INFERA_DENSE16_STAGED,
//   INFERA_DENSE16G_MIN_M, INFERA_SOFTMAX_ROWS, INFERA_POOL_FAST, INFERA_CHAIN_WAVES, INFERA_MLP3_VARIANT (PROBES builds), INFERA_CONV_PROBE.

void log_msg(int level, const std::string &msg);  // config.rs:200-207 `log!`

// ---------------------------------------------------------------------------------------------
// JSON emitters (serde_json compact form: no spaces, keys of json!{} objects sorted)
// ---------------------------------------------------------------------------------------------
std::string json_escape(const std::string &s);
inline std::string json_str(const std::string &s) { return "\"" + json_escape(s) + "\""; }
template <typename T>
std::string json_int_array(const std::vector<T> &v) {
  std::ostringstream o;
  o << "[";
  for (size_t i = 0; i < v.size(); i++) o << (i ? "," : "") << v[i];
  o << "]";
  return o.str();
}
std::string json_str_array(const std::vector<std::string> &v);

}  // namespace infera_hip
