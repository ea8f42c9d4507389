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
                                      //   host-path chunk (default: DESIGN.md 4, set by the round-4 A/B on CPU time per chunk)
  int max_inflight_total;             // INFERA_MAX_INFLIGHT_TOTAL=n  host-ABI calls the PROCESS admits between first H2D and sync over all
                                      //   GPUs (0 = no process-wide limit; the per-GPU limit is INFERA_MAX_INFLIGHT)
  bool host_zero_copy;                // INFERA_HOST_ZERO_COPY=0|1 (default 1)  infera_predict_columns chunks whose column runs all lie in host memory
                                      //   registered with infera_hip_register_host_memory are read in place by the GPU (no CPU gather, no H2D copy)
  bool zero_copy_rect;                // INFERA_ZERO_COPY_RECT=0|1 (default 1)  a zero-copy chunk whose runs are FLOAT at one stride inside one pinned block is fetched by ONE
                                      //   2-D copy (two or three callers alternate on the one-at-a-time 2-D copy engine path: 64 / 80 M rows/s against the pulling kernel's 52 / 65); 0: always the pulling kernel
  int zero_copy_rect_inflight;        // INFERA_ZERO_COPY_RECT_INFLIGHT=n (default 3; 0 = no limit)  2-D copies a GPU runs at a time; chunks beyond that take the pulling kernel
  int zero_copy_max_inflight;         // INFERA_ZERO_COPY_MAX_INFLIGHT=n (default 0 = no limit)  zero-copy fetches a GPU runs at a time; chunks beyond that are STAGED
                                      //   (round 4 shipped 4: in-place fetches stopped at 80 M rows/s per GPU then -- host_path.cpp, g_zc_fetches)
  bool numa_slots;                    // INFERA_NUMA_SLOTS=0|1 (default 1)  caller threads prefer the device slots on their own NUMA node (bounded by load)
  long long host_direct_in_bytes;     // INFERA_HOST_DIRECT_IN=<bytes>  chunks up to this size are read from pinned memory by the first kernel itself
                                      //   (default 131072; 0 = always H2D).  On a quiet GPU the effective limit is higher: x2 with at most four
                                      //   host-ABI calls in flight, x4 (512 KB) with at most two
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
  bool conv_bf16x6; // INFERA_PRECISION unset | bf16x6 (default): the tiled convolutions and the 7x7/2 stem on the bf16 matrix cores, every fp32 operand cut
                    //   EXACTLY into three bf16 parts, six partial products per product, fp32 accumulate (conv_split.hip) -- no scales, no precondition
                    //   on the data.  INFERA_PRECISION=fp32: the exact-fp32 matrix instruction instead (conv.hip's tiled / weight-stationary kernels)
  bool conv_fold_shortcut;  // INFERA_CONV_FOLD_SHORTCUT=0|1 (default 1)  a ResNet block's 1x1 projection shortcut as extra K stages of the block's second convolution
  static ScheduleKnobs read();
};
// Read per launch inside the kernel launchers, for the bit-identity TESTS only (no effect on results; defaults are the shipped paths):
//   INFERA_CONV_WS (0 tiled kernel only | 1 default | 2 force the weight-stationary kernel), INFERA_CONV_TAIL_SPLIT, INFERA_STEM_POOL2,
//   INFERA_STEM_SPLIT (0: the exact-fp32 stem kernels under a default plan); INFERA_CONV_LANES=1 (hip/exec.cpp: long convolutional passes
//   as ONE lane -- counter passes serialise kernels, so tools/profile_bench.sh takes its PMC passes this way; read once).

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
