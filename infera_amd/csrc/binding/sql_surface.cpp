// sql_surface.cpp -- see sql_surface.h.  Each function cites the reference function it mirrors.
#include "sql_surface.h"

#include <atomic>
#include <chrono>
#include <sched.h>
#include <sys/resource.h>
#include <cstdio>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <charconv>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/infera_hip.h"

namespace {

using namespace infera;

// DuckDB reports InvalidInputException as "Invalid Input Error: <msg>" (the prefix the reference's
// sqllogictests assert, e.g. test/sql/test_edge_cases.test:27-30).
struct InvalidInput : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct Chunk {
  const InferaSqlVector *v;
  size_t ncols, rows;
  size_t size() const { return rows; }
  size_t ColumnCount() const { return ncols; }
};

bool is_null(const InferaSqlVector &v, size_t row) {
  if (!v.validity) return false;
  const size_t r = v.is_constant ? 0 : row;
  return !((v.validity[r >> 6] >> (r & 63)) & 1);
}

// any NULL among the first `rows` entries: whole validity words at a time (a chunk is 128 columns x 2048 rows; a
// per-cell test here cost more than the gather itself -- 56 of 196 us per chunk)
bool any_null(const InferaSqlVector &v, size_t rows) {
  if (!v.validity || rows == 0) return false;
  if (v.is_constant) return !(v.validity[0] & 1);
  const size_t full = rows >> 6, rest = rows & 63;
  for (size_t w = 0; w < full; w++)
    if (v.validity[w] != ~uint64_t(0)) return true;
  return rest && (v.validity[full] & ((uint64_t(1) << rest) - 1)) != ((uint64_t(1) << rest) - 1);
}

std::string get_string(const InferaSqlVector &v, size_t row) {
  const size_t r = v.is_constant ? 0 : row;
  auto ptrs = static_cast<const uint8_t *const *>(v.data);
  const size_t len = v.lens ? size_t(v.lens[r]) : std::strlen(reinterpret_cast<const char *>(ptrs[r]));
  return std::string(reinterpret_cast<const char *>(ptrs[r]), len);
}

const char *type_name(int t) {
  static const char *n[] = {"VARCHAR", "FLOAT", "DOUBLE", "INTEGER", "BIGINT", "BLOB", "BOOLEAN", "FLOAT[]"};
  return (t >= 0 && t < 8) ? n[t] : "UNKNOWN";
}

std::string last_error() {  // GetInferaError, infera_extension.cpp:52-55
  const char *e = infera_last_error();
  return e ? std::string(e) : std::string("unknown error");
}

void set_constant_bool(InferaSqlResult *out, bool v) {
  out->type = INFERA_SQL_BOOLEAN;
  out->is_constant = 1;
  out->boolean = static_cast<uint8_t *>(std::malloc(1));
  out->boolean[0] = v;
}

void set_constant_string(InferaSqlResult *out, const std::string &s) {
  out->type = INFERA_SQL_VARCHAR;
  out->is_constant = 1;
  out->strings = static_cast<char **>(std::malloc(sizeof(char *)));
  out->strings[0] = strdup(s.c_str());
}

void set_constant_null(InferaSqlResult *out, int type) {
  out->type = type;
  out->is_constant = 1;
  out->validity = static_cast<uint64_t *>(std::calloc(1, sizeof(uint64_t)));
}

// ValidateAndGetModelName, infera_extension.cpp:239-248 (name from row 0 only)
std::string validate_and_get_model_name(const Chunk &args, const std::string &func) {
  if (args.ColumnCount() < 2) throw InvalidInput(func + "(model_name, feature1, ...) requires at least 2 arguments");
  if (is_null(args.v[0], 0)) throw InvalidInput("Model name cannot be NULL");
  return get_string(args.v[0], 0);
}

// ExtractFeatures (infera_extension.cpp:199-227) without the per-cell boxing: validity masks are
// scanned per column, the typed flat pointers go to infera_predict_columns, which gathers them into
// the row-major f32 buffer with static_cast<float> per type.
InferaInferenceResult predict_chunk(const Chunk &args, const std::string &model) {
  const size_t F = args.ColumnCount() - 1, rows = args.size();
  std::vector<InferaColumn> cols(F);
  for (size_t c = 0; c < F; c++) {
    const InferaSqlVector &v = args.v[c + 1];
    if (any_null(v, rows)) throw InvalidInput("Feature values cannot be NULL");  // :207-209
    switch (v.type) {
      case INFERA_SQL_FLOAT: cols[c].type = INFERA_COL_FLOAT; break;
      case INFERA_SQL_DOUBLE: cols[c].type = INFERA_COL_DOUBLE; break;
      case INFERA_SQL_INTEGER: cols[c].type = INFERA_COL_INTEGER; break;
      case INFERA_SQL_BIGINT: cols[c].type = INFERA_COL_BIGINT; break;
      default: throw InvalidInput(std::string("Unsupported feature type: ") + type_name(v.type));  // :222
    }
    cols[c].data = v.data;
    cols[c].validity = nullptr;
    cols[c].is_constant = v.is_constant;
  }
  InferaInferenceResult res = infera_predict_columns(model.c_str(), cols.data(), F, rows);
  if (res.status != 0) {  // :271-274
    infera_free_result(res);
    throw InvalidInput("Inference failed for model '" + model + "': " + last_error());
  }
  return res;
}

std::string fmt_shape_mismatch(size_t batch, size_t r, size_t c) {  // StringUtil::Format, :276
  char buf[160];
  std::snprintf(buf, sizeof buf, "Model output shape mismatch. Expected (%d, 1), but got (%d, %d).", int(batch), int(r), int(c));
  return buf;
}
std::string fmt_row_mismatch(size_t batch, size_t r) {  // :398, :446
  char buf[160];
  std::snprintf(buf, sizeof buf, "Model output row count mismatch. Expected %d, but got %d.", int(batch), int(r));
  return buf;
}

// Predict, infera_extension.cpp:260-286
void Predict(const Chunk &args, InferaSqlResult *out) {
  out->type = INFERA_SQL_FLOAT;
  if (args.size() == 0) return;
  const std::string model = validate_and_get_model_name(args, "infera_predict");
  const size_t batch = args.size();
  InferaInferenceResult res = predict_chunk(args, model);
  if (res.rows != batch || res.cols != 1) {
    std::string msg = fmt_shape_mismatch(batch, res.rows, res.cols);
    infera_free_result(res);
    throw InvalidInput(msg);
  }
  out->f32 = static_cast<float *>(std::malloc(batch * sizeof(float)));
  std::memcpy(out->f32, res.data, batch * sizeof(float));  // :280-284
  infera_free_result(res);
}

// PredictMulti, infera_extension.cpp:382-418 -- "[a,b,c]" via ostream << float (%g, 6 digits)
void PredictMulti(const Chunk &args, InferaSqlResult *out) {
  out->type = INFERA_SQL_VARCHAR;
  if (args.size() == 0) return;
  const std::string model = validate_and_get_model_name(args, "infera_predict_multi");
  const size_t batch = args.size();
  InferaInferenceResult res = predict_chunk(args, model);
  if (res.rows != batch) {
    std::string msg = fmt_row_mismatch(batch, res.rows);
    infera_free_result(res);
    throw InvalidInput(msg);
  }
  out->strings = static_cast<char **>(std::calloc(batch, sizeof(char *)));
  // std::to_chars(general, 6) prints exactly what `ostream << float` prints (%g, 6 significant digits; checked
  // on 3M values incl. every class of bit pattern) at a quarter of the cost and without a stream per row
  std::vector<char> line(res.cols * 17 + 3);
  for (size_t r = 0; r < batch; r++) {
    char *p = line.data();
    *p++ = '[';
    for (size_t c = 0; c < res.cols; c++) {
      if (c) *p++ = ',';
      p = std::to_chars(p, p + 16, res.data[r * res.cols + c], std::chars_format::general, 6).ptr;
    }
    *p++ = ']';
    const size_t len = size_t(p - line.data());
    char *str = static_cast<char *>(std::malloc(len + 1));
    std::memcpy(str, line.data(), len);
    str[len] = 0;
    out->strings[r] = str;
  }
  infera_free_result(res);
}

// PredictMultiList, infera_extension.cpp:430-462 -- list children written in one block
void PredictMultiList(const Chunk &args, InferaSqlResult *out, const char *fname) {
  out->type = INFERA_SQL_LIST_FLOAT;
  if (args.size() == 0) return;
  const std::string model = validate_and_get_model_name(args, fname);
  const size_t batch = args.size();
  InferaInferenceResult res = predict_chunk(args, model);
  if (res.rows != batch) {
    std::string msg = fmt_row_mismatch(batch, res.rows);
    infera_free_result(res);
    throw InvalidInput(msg);
  }
  out->list_offsets = static_cast<uint64_t *>(std::malloc((batch + 1) * sizeof(uint64_t)));
  for (size_t r = 0; r <= batch; r++) out->list_offsets[r] = r * res.cols;
  out->list_values = static_cast<float *>(std::malloc(std::max<size_t>(res.len, 1) * sizeof(float)));
  std::memcpy(out->list_values, res.data, res.len * sizeof(float));
  infera_free_result(res);
}

// PredictFromBlob, infera_extension.cpp:297-328
void PredictFromBlob(const Chunk &args, InferaSqlResult *out) {
  out->type = INFERA_SQL_LIST_FLOAT;
  if (args.ColumnCount() != 2) throw InvalidInput("infera_predict_from_blob(model_name, input_blob) requires 2 arguments");
  const size_t n = args.size();
  if (n == 0) return;
  std::vector<std::vector<float>> lists(n);
  std::vector<bool> valid(n, false);
  std::vector<size_t> live;
  for (size_t i = 0; i < n; i++)
    if (!is_null(args.v[0], i) && !is_null(args.v[1], i)) live.push_back(i);  // :306-309 NULL in -> NULL out
  auto blob_ptr = [&](size_t i) { return static_cast<const uint8_t *const *>(args.v[1].data)[args.v[1].is_constant ? 0 : i]; };
  auto blob_len = [&](size_t i) { return size_t(args.v[1].lens[args.v[1].is_constant ? 0 : i]); };
  // One batched FFI call when the whole chunk uses one model and every blob holds one sample
  // (ROADMAP.md:43 "single FFI call"); anything else takes the reference's per-row route, which
  // also reproduces its per-row error texts.
  bool batched = false;
  if (live.size() > 1) {
    const std::string first = get_string(args.v[0], live[0]);
    bool same = true;
    for (size_t i : live) same = same && get_string(args.v[0], i) == first;
    if (same) {
      std::vector<const uint8_t *> ptrs;
      std::vector<uintptr_t> lens;
      for (size_t i : live) {
        ptrs.push_back(blob_ptr(i));
        lens.push_back(blob_len(i));
      }
      InferaInferenceResult res = infera_predict_from_blob_batch(first.c_str(), ptrs.data(), lens.data(), live.size());
      if (res.status == 0 && res.rows == live.size()) {
        for (size_t k = 0; k < live.size(); k++) {
          lists[live[k]].assign(res.data + k * res.cols, res.data + (k + 1) * res.cols);
          valid[live[k]] = true;
        }
        batched = true;
      }
      infera_free_result(res);
    }
  }
  if (!batched) {
    for (size_t i : live) {
      const std::string model = get_string(args.v[0], i);
      InferaInferenceResult res = infera_predict_from_blob(model.c_str(), blob_ptr(i), blob_len(i));
      if (res.status != 0) {
        infera_free_result(res);
        throw InvalidInput("Inference failed for model '" + model + "': " + last_error());  // :315-318
      }
      lists[i].assign(res.data, res.data + res.len);  // all res.len elements (:319-325)
      valid[i] = true;
      infera_free_result(res);
    }
  }
  out->list_offsets = static_cast<uint64_t *>(std::malloc((n + 1) * sizeof(uint64_t)));
  size_t total = 0;
  for (size_t i = 0; i < n; i++) {
    out->list_offsets[i] = total;
    total += lists[i].size();
  }
  out->list_offsets[n] = total;
  out->list_values = static_cast<float *>(std::malloc(std::max<size_t>(total, 1) * sizeof(float)));
  for (size_t i = 0; i < n; i++)
    if (!lists[i].empty()) std::memcpy(out->list_values + out->list_offsets[i], lists[i].data(), lists[i].size() * sizeof(float));
  if (live.size() != n) {
    out->validity = static_cast<uint64_t *>(std::calloc((n + 63) / 64, sizeof(uint64_t)));
    for (size_t i = 0; i < n; i++)
      if (valid[i]) out->validity[i >> 6] |= uint64_t(1) << (i & 63);
  }
}

// LoadModel, infera_extension.cpp:133-156
void LoadModel(const Chunk &args, InferaSqlResult *out) {
  if (args.ColumnCount() != 2) throw InvalidInput("infera_load_model(model_name, path) expects exactly 2 arguments");
  out->type = INFERA_SQL_BOOLEAN;
  if (args.size() == 0) return;
  if (is_null(args.v[0], 0) || is_null(args.v[1], 0)) throw InvalidInput("Model name and path cannot be NULL");
  const std::string name = get_string(args.v[0], 0), path = get_string(args.v[1], 0);
  if (name.empty()) throw InvalidInput("Model name cannot be empty");
  if (infera_load_model(name.c_str(), path.c_str()) != 0) throw InvalidInput("Failed to load model '" + name + "': " + last_error());
  set_constant_bool(out, true);
}

// UnloadModel, infera_extension.cpp:167-188 (model-not-found is a benign idempotent success)
void UnloadModel(const Chunk &args, InferaSqlResult *out) {
  if (args.ColumnCount() != 1) throw InvalidInput("infera_unload_model(model_name) expects exactly 1 argument");
  out->type = INFERA_SQL_BOOLEAN;
  if (args.size() == 0) return;
  if (is_null(args.v[0], 0)) throw InvalidInput("Model name cannot be NULL");
  const std::string name = get_string(args.v[0], 0);
  if (infera_unload_model(name.c_str()) != 0) {
    std::string err = last_error();
    if (err.rfind("Model not found:", 0) != 0) throw InvalidInput("Failed to unload model '" + name + "': " + err);
  }
  set_constant_bool(out, true);
}

std::string take(char *p, const char *dflt = "") {
  std::string s = p ? std::string(p) : std::string(dflt);
  infera_free(p);
  return s;
}

// GetModelInfo, infera_extension.cpp:473-497
void GetModelInfo(const Chunk &args, InferaSqlResult *out) {
  if (args.ColumnCount() != 1) throw InvalidInput("infera_get_model_info(model_name) expects exactly 1 argument");
  out->type = INFERA_SQL_VARCHAR;
  if (args.size() == 0) return;
  if (is_null(args.v[0], 0)) throw InvalidInput("Model name cannot be NULL");
  const std::string name = get_string(args.v[0], 0);
  const std::string js = take(infera_get_model_info(name.c_str()));
  if (js.empty() || js.find("\"error\"") != std::string::npos) throw InvalidInput("Failed to get info for model '" + name + "'");
  set_constant_string(out, js);
}

// IsModelLoaded, infera_extension.cpp:349-370 (substring search for the quoted name)
void IsModelLoaded(const Chunk &args, InferaSqlResult *out) {
  if (args.ColumnCount() != 1) throw InvalidInput("infera_is_model_loaded(model_name) expects exactly 1 argument");
  out->type = INFERA_SQL_BOOLEAN;
  if (args.size() == 0) return;
  if (is_null(args.v[0], 0)) throw InvalidInput("Model name cannot be NULL");
  const std::string js = take(infera_get_loaded_models());
  set_constant_bool(out, js.find("\"" + get_string(args.v[0], 0) + "\"") != std::string::npos);
}

// SetAutoloadDir, infera_extension.cpp:88-104
void SetAutoloadDir(const Chunk &args, InferaSqlResult *out) {
  if (args.ColumnCount() != 1) throw InvalidInput("infera_set_autoload_dir(path) expects exactly 1 argument");
  out->type = INFERA_SQL_VARCHAR;
  if (args.size() == 0) return;
  if (is_null(args.v[0], 0)) throw InvalidInput("Path cannot be NULL");
  set_constant_string(out, take(infera_set_autoload_dir(get_string(args.v[0], 0).c_str())));
}

struct FnInfo {
  const char *name;
  int min_args, max_args;
  const char *returns;
  bool is_volatile;
  bool default_null_handling;  // constant-NULL argument -> NULL result without calling
};

// LoadInternal, infera_extension.cpp:546-592 (+ infera_predict_array, + feature cap raised)
const FnInfo kFunctions[] = {
    {"infera_load_model", 2, 2, "BOOLEAN", true, true},
    {"infera_unload_model", 1, 1, "BOOLEAN", true, true},
    {"infera_predict", 2, 1 + INFERA_SQL_MAX_FEATURES, "FLOAT", true, true},
    {"infera_predict_multi", 2, 1 + INFERA_SQL_MAX_FEATURES, "VARCHAR", true, true},
    {"infera_predict_multi_list", 2, 1 + INFERA_SQL_MAX_FEATURES, "FLOAT[]", true, true},
    {"infera_predict_array", 2, 1 + INFERA_SQL_MAX_FEATURES, "FLOAT[]", true, true},
    {"infera_predict_from_blob", 2, 2, "FLOAT[]", true, true},
    {"infera_get_loaded_models", 0, 0, "VARCHAR", true, true},
    {"infera_get_model_info", 1, 1, "VARCHAR", true, true},
    {"infera_get_version", 0, 0, "VARCHAR", false, true},
    {"infera_set_autoload_dir", 1, 1, "VARCHAR", true, true},
    {"infera_is_model_loaded", 1, 1, "BOOLEAN", true, true},
    {"infera_clear_cache", 0, 0, "BOOLEAN", true, true},
    {"infera_get_cache_info", 0, 0, "VARCHAR", true, true},
};

int result_type_of(const std::string &r) {
  if (r == "FLOAT") return INFERA_SQL_FLOAT;
  if (r == "BOOLEAN") return INFERA_SQL_BOOLEAN;
  if (r == "VARCHAR") return INFERA_SQL_VARCHAR;
  return INFERA_SQL_LIST_FLOAT;
}

}  // namespace

extern "C" {

int32_t infera_sql_call(const char *function, const InferaSqlVector *argv, uintptr_t nargs, uintptr_t rows, InferaSqlResult *out) {
  std::memset(out, 0, sizeof *out);
  out->rows = rows;
  try {
    if (!function) throw InvalidInput("function name is NULL");
    const std::string fn = function;
    const FnInfo *info = nullptr;
    for (const auto &f : kFunctions)
      if (fn == f.name) info = &f;
    if (!info) throw std::runtime_error("Catalog Error: Scalar Function with name " + fn + " does not exist!");
    if (int(nargs) < info->min_args || int(nargs) > info->max_args)
      throw std::runtime_error("Binder Error: No function matches the given name and argument types '" + fn + "' with " +
                               std::to_string(nargs) + " arguments");
    if (rows > INFERA_SQL_VECTOR_SIZE) throw InvalidInput("chunk larger than STANDARD_VECTOR_SIZE");
    Chunk args{argv, size_t(nargs), size_t(rows)};
    // DuckDB's default NULL handling: a constant NULL argument short-circuits to a constant NULL
    // result; the function body is never entered (test/sql/test_edge_cases.test:38-42).
    if (info->default_null_handling && rows > 0)
      for (size_t c = 0; c < nargs; c++)
        if (argv[c].is_constant && is_null(argv[c], 0)) {
          set_constant_null(out, result_type_of(info->returns));
          return 0;
        }
    if (fn == "infera_predict") Predict(args, out);
    else if (fn == "infera_predict_multi") PredictMulti(args, out);
    else if (fn == "infera_predict_multi_list" || fn == "infera_predict_array") PredictMultiList(args, out, function);
    else if (fn == "infera_predict_from_blob") PredictFromBlob(args, out);
    else if (fn == "infera_load_model") LoadModel(args, out);
    else if (fn == "infera_unload_model") UnloadModel(args, out);
    else if (fn == "infera_get_model_info") GetModelInfo(args, out);
    else if (fn == "infera_is_model_loaded") IsModelLoaded(args, out);
    else if (fn == "infera_set_autoload_dir") SetAutoloadDir(args, out);
    else if (fn == "infera_get_loaded_models") set_constant_string(out, take(infera_get_loaded_models(), "[]"));  // :339-347
    else if (fn == "infera_get_version") set_constant_string(out, take(infera_get_version()));                    // :115-121
    else if (fn == "infera_get_cache_info") set_constant_string(out, take(infera_get_cache_info()));              // :530-536
    else if (fn == "infera_clear_cache") {                                                                         // :508-518
      if (infera_clear_cache() != 0) throw InvalidInput("Failed to clear cache: " + last_error());
      set_constant_bool(out, true);
    }
    return 0;
  } catch (const InvalidInput &e) {
    infera_sql_free_result(out);
    out->status = -1;
    out->error = strdup((std::string("Invalid Input Error: ") + e.what()).c_str());
  } catch (const std::exception &e) {
    infera_sql_free_result(out);
    out->status = -1;
    out->error = strdup(e.what());
  }
  return -1;
}

void infera_sql_free_result(InferaSqlResult *r) {
  if (!r) return;
  std::free(r->error);
  std::free(r->f32);
  std::free(r->boolean);
  if (r->strings) {
    const uint64_t n = r->is_constant ? 1 : r->rows;
    for (uint64_t i = 0; i < n; i++) std::free(r->strings[i]);
    std::free(r->strings);
  }
  std::free(r->list_offsets);
  std::free(r->list_values);
  std::free(r->validity);
  const uint64_t rows = r->rows;
  std::memset(r, 0, sizeof *r);
  r->rows = rows;
}

char *infera_sql_list_functions(void) {
  std::string o = "[";
  bool first = true;
  for (const auto &f : kFunctions) {
    if (!first) o += ",";
    first = false;
    o += std::string("{\"name\":\"") + f.name + "\",\"min_args\":" + std::to_string(f.min_args) + ",\"max_args\":" +
         std::to_string(f.max_args) + ",\"returns\":\"" + f.returns + "\",\"volatile\":" + (f.is_volatile ? "true" : "false") + "}";
  }
  o += "]";
  return strdup(o.c_str());
}

namespace {
uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
}  // namespace

double infera_sql_bench_scan(const char *function, const char *model, uint64_t rows, uint32_t ncols, int32_t threads,
                             int32_t pool_chunks, uint64_t seed, double *checksum, char *err, uint64_t errlen) {
  if (threads < 1) threads = 1;
  if (pool_chunks < 1) pool_chunks = 1;
  const size_t CH = INFERA_SQL_VECTOR_SIZE;
  const uint64_t nchunks = (rows + CH - 1) / CH;
  std::atomic<uint64_t> next{0};
  std::mutex mu;
  std::string first_error;
  double total = 0.0;
  std::vector<std::vector<float>> pools((size_t)threads);  // [thread] -> pool_chunks x ncols x CH, column-major per chunk
  for (int t = 0; t < threads; t++) {
    pools[size_t(t)].resize(size_t(pool_chunks) * ncols * CH);
    for (int p = 0; p < pool_chunks; p++)
      for (uint32_t c = 0; c < ncols; c++)
        for (size_t r = 0; r < CH; r++) {
          const uint64_t row = (uint64_t(t) * uint64_t(pool_chunks) + uint64_t(p)) * CH + r;
          const uint64_t u = splitmix64(seed ^ (row * ncols + c));
          pools[size_t(t)][(size_t(p) * ncols + c) * CH + r] = float(u >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
        }
  }
  const std::string fn = function ? function : "infera_predict";
  auto worker = [&](int t) {
    std::vector<InferaSqlVector> args(ncols + 1);
    const uint8_t *name_ptr = reinterpret_cast<const uint8_t *>(model);
    uint64_t name_len = std::strlen(model);
    args[0].type = INFERA_SQL_VARCHAR;
    args[0].is_constant = 1;
    args[0].data = &name_ptr;
    args[0].lens = &name_len;
    args[0].validity = nullptr;
    double local = 0.0;
    uint64_t k = 0;
    for (;;) {
      const uint64_t c = next.fetch_add(1, std::memory_order_relaxed);
      if (c >= nchunks) break;
      const size_t nr = size_t(std::min<uint64_t>(CH, rows - c * CH));
      const float *base = pools[size_t(t)].data() + size_t(k++ % uint64_t(pool_chunks)) * ncols * CH;
      for (uint32_t j = 0; j < ncols; j++) {
        args[j + 1].type = INFERA_SQL_FLOAT;
        args[j + 1].is_constant = 0;
        args[j + 1].data = base + size_t(j) * CH;
        args[j + 1].lens = nullptr;
        args[j + 1].validity = nullptr;
      }
      InferaSqlResult res;
      if (infera_sql_call(fn.c_str(), args.data(), ncols + 1, nr, &res) != 0) {
        std::lock_guard<std::mutex> lk(mu);
        if (first_error.empty()) first_error = res.error ? res.error : "unknown error";
        infera_sql_free_result(&res);
        next.store(nchunks);
        break;
      }
      if (res.f32)
        for (size_t i = 0; i < nr; i++) local += double(res.f32[i]);
      else if (res.list_offsets)
        for (uint64_t i = 0; i < res.list_offsets[nr]; i++) local += double(res.list_values[i]);
      infera_sql_free_result(&res);
    }
    std::lock_guard<std::mutex> lk(mu);
    total += local;
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(worker, t);
  for (auto &x : th) x.join();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (checksum) *checksum = total;
  if (!first_error.empty()) {
    if (err && errlen) std::snprintf(err, size_t(errlen), "%s", first_error.c_str());
    return -1.0;
  }
  return sec;
}

// ---- table scan over a MATERIALISED columnar table (the measurement of SURVEY.md 8d) --------------------------
// Layout = DuckDB's storage shape: row groups of INFERA_SQL_ROW_GROUP rows, inside a group one contiguous run per
// column.  Value (row, col) sits at  table[g*RG*ncols + col*rows_in_group(g) + (row - g*RG)].

uint64_t infera_sql_table_floats(uint64_t rows, uint32_t ncols) { return rows * uint64_t(ncols); }

void infera_sql_synth_table(float *table, uint64_t seed, uint64_t rows, uint32_t ncols, int32_t threads) {
  if (threads < 1) threads = 1;
  const uint64_t RG = INFERA_SQL_ROW_GROUP;
  const uint64_t ngroups = (rows + RG - 1) / RG;
  std::atomic<uint64_t> next{0};
  auto worker = [&] {
    for (;;) {
      const uint64_t task = next.fetch_add(1, std::memory_order_relaxed);  // one (group, column) run per task
      if (task >= ngroups * ncols) break;
      const uint64_t g = task / ncols, c = task % ncols;
      const uint64_t r0 = g * RG, gr = std::min<uint64_t>(RG, rows - r0);
      float *dst = table + r0 * ncols + c * gr;
      for (uint64_t r = 0; r < gr; r++) {
        const uint64_t u = splitmix64(seed ^ ((r0 + r) * ncols + c));
        dst[r] = float(u >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(worker);
  for (auto &x : th) x.join();
}

namespace {
// process CPU time (user + system, all threads) -- what a cgroup CPU quota meters
double process_cpu_seconds(double *sys_out = nullptr) {
  rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  const double sys = double(ru.ru_stime.tv_sec) + double(ru.ru_stime.tv_usec) * 1e-6;
  if (sys_out) *sys_out = sys;
  return double(ru.ru_utime.tv_sec) + double(ru.ru_utime.tv_usec) * 1e-6 + sys;
}
double g_bench_cpu_s = 0.0, g_bench_sys_s = 0.0, g_bench_wall_s = 0.0;
std::atomic<uint64_t> g_bench_call_ns{0}, g_bench_thread_ns{0};
inline uint64_t bench_now_ns() {
  return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
}  // namespace

namespace {
// sum of a result block with eight independent partial sums (a dependent chain of 20,480 double adds per 10-class
// chunk cost 50 us -- more than the chunk's H2D copy)
double sum_block(const float *v, size_t n) {
  double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  size_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int k = 0; k < 8; k++) p[k] += double(v[i + k]);
  for (; i < n; i++) p[0] += double(v[i]);
  return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}
}  // namespace

void infera_sql_bench_last_cpu(double *cpu_seconds, double *sys_seconds, double *wall_seconds) {
  if (cpu_seconds) *cpu_seconds = g_bench_cpu_s;
  if (sys_seconds) *sys_seconds = g_bench_sys_s;
  if (wall_seconds) *wall_seconds = g_bench_wall_s;
}

void infera_sql_bench_last_times(uint64_t *call_ns, uint64_t *thread_ns) {
  if (call_ns) *call_ns = g_bench_call_ns.load();
  if (thread_ns) *thread_ns = g_bench_thread_ns.load();
}

int32_t infera_sql_bench_scan_table(const char *function, const char *model, const float *table, uint64_t rows, uint32_t ncols,
                                    int32_t threads, int32_t reps, double *secs, double *checksum, char *err, uint64_t errlen) {
  return infera_sql_bench_scan_table_typed(function, model, table, INFERA_SQL_FLOAT, rows, ncols, threads, reps, secs, checksum, err, errlen);
}

void infera_sql_synth_table_f64(double *table, uint64_t seed, uint64_t rows, uint32_t ncols, int32_t threads) {
  if (threads < 1) threads = 1;
  const uint64_t RG = INFERA_SQL_ROW_GROUP, ngroups = (rows + RG - 1) / RG;
  std::atomic<uint64_t> next{0};
  auto worker = [&] {
    for (;;) {
      const uint64_t task = next.fetch_add(1, std::memory_order_relaxed);
      if (task >= ngroups * ncols) break;
      const uint64_t g = task / ncols, c = task % ncols, r0 = g * RG, gr = std::min<uint64_t>(RG, rows - r0);
      double *dst = table + r0 * ncols + c * gr;
      for (uint64_t r = 0; r < gr; r++) {
        const uint64_t u = splitmix64(seed ^ ((r0 + r) * ncols + c));
        dst[r] = double(float(u >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f);  // the same values as the FLOAT table, widened
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(worker);
  for (auto &x : th) x.join();
}

int32_t infera_sql_bench_scan_table_typed(const char *function, const char *model, const void *table_v, int32_t elem_type, uint64_t rows,
                                          uint32_t ncols, int32_t threads, int32_t reps, double *secs, double *checksum, char *err,
                                          uint64_t errlen) {
  if (elem_type != INFERA_SQL_FLOAT && elem_type != INFERA_SQL_DOUBLE) {
    if (err && errlen) std::snprintf(err, size_t(errlen), "table element type must be FLOAT or DOUBLE");
    return -1;
  }
  const size_t esz = elem_type == INFERA_SQL_DOUBLE ? 8 : 4;
  const uint8_t *table = static_cast<const uint8_t *>(table_v);
  if (threads < 1) threads = 1;
  const size_t CH = INFERA_SQL_VECTOR_SIZE;
  const uint64_t RG = INFERA_SQL_ROW_GROUP;
  static_assert(INFERA_SQL_ROW_GROUP % INFERA_SQL_VECTOR_SIZE == 0, "chunks never straddle a row group");
  const uint64_t nchunks = (rows + CH - 1) / CH;
  const std::string fn = function ? function : "infera_predict";
  std::string first_error;
  std::mutex mu;
  for (int rep = 0; rep < reps; rep++) {
    std::atomic<uint64_t> next{0};
    double total = 0.0;
    std::atomic<int> worker_id{0};
    auto worker = [&] {
      // INFERA_BENCH_PIN=spread|pack: EXPERIMENT ONLY -- pins scan worker t to one CPU of the process's affinity mask (spread: every
      // 8th CPU = one per CCD first; pack: consecutive CPUs).  DuckDB does not pin its workers; this separates what thread
      // migration costs the gather from what the memory system costs it.
      if (const char *pin = std::getenv("INFERA_BENCH_PIN")) {
        cpu_set_t mask;
        if (sched_getaffinity(0, sizeof mask, &mask) == 0) {
          std::vector<int> cpus;
          for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &mask)) cpus.push_back(c);
          const int me = worker_id.fetch_add(1), n = int(cpus.size());
          if (n > 0) {
            const int idx = pin[0] == 's' ? int((int64_t(me) * 8) % n + (int64_t(me) * 8) / n) % n : me % n;
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[size_t(idx)], &one);
            (void)sched_setaffinity(0, sizeof one, &one);
          }
        }
      }
      std::vector<InferaSqlVector> args(ncols + 1);
      const uint8_t *name_ptr = reinterpret_cast<const uint8_t *>(model);
      uint64_t name_len = std::strlen(model);
      args[0] = InferaSqlVector{INFERA_SQL_VARCHAR, 1, &name_ptr, &name_len, nullptr};
      double local = 0.0;
      uint64_t in_call = 0;
      const uint64_t t_thread0 = bench_now_ns();
      for (;;) {
        const uint64_t c = next.fetch_add(1, std::memory_order_relaxed);
        if (c >= nchunks) break;
        const uint64_t row0 = c * CH, g = row0 / RG, g0 = g * RG, gr = std::min<uint64_t>(RG, rows - g0);
        const size_t nr = size_t(std::min<uint64_t>(CH, rows - row0));
        const uint8_t *base = table + (g0 * ncols + (row0 - g0)) * esz;
        for (uint32_t j = 0; j < ncols; j++) args[j + 1] = InferaSqlVector{elem_type, 0, base + uint64_t(j) * gr * esz, nullptr, nullptr};
        InferaSqlResult res;
        const uint64_t t_c0 = bench_now_ns();
        const int32_t rc = infera_sql_call(fn.c_str(), args.data(), ncols + 1, nr, &res);
        in_call += bench_now_ns() - t_c0;
        if (rc != 0) {
          std::lock_guard<std::mutex> lk(mu);
          if (first_error.empty()) first_error = res.error ? res.error : "unknown error";
          infera_sql_free_result(&res);
          next.store(nchunks);
          break;
        }
        // the consumer of the result vector (an aggregate above the scan) touches every element once
        if (res.f32) local += sum_block(res.f32, nr);
        else if (res.list_offsets) local += sum_block(res.list_values, size_t(res.list_offsets[nr]));
        infera_sql_free_result(&res);
      }
      g_bench_call_ns.fetch_add(in_call);
      g_bench_thread_ns.fetch_add(bench_now_ns() - t_thread0);
      std::lock_guard<std::mutex> lk(mu);
      total += local;
    };
    if (rep == 0) {
      g_bench_call_ns = 0;
      g_bench_thread_ns = 0;
      g_bench_cpu_s = g_bench_sys_s = g_bench_wall_s = 0.0;
    }
    double sys0 = 0.0, sys1 = 0.0;
    const double cpu0 = process_cpu_seconds(&sys0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker);
    for (auto &x : th) x.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_bench_cpu_s += process_cpu_seconds(&sys1) - cpu0;  // every thread of the process: workers, HIP runtime helpers, interrupts' bottom halves charged to us
    g_bench_sys_s += sys1 - sys0;
    g_bench_wall_s += wall;
    if (secs) secs[rep] = wall;
    if (checksum) *checksum = total;
    if (!first_error.empty()) {
      if (err && errlen) std::snprintf(err, size_t(errlen), "%s", first_error.c_str());
      return -1;
    }
  }
  return 0;
}


// The BLOB path's scan (BASELINE config C5): `threads` workers pull 2048-row chunks of an image table whose BLOBs live in
// host memory (`nblobs` blobs of `blob_bytes` each, row r uses blob r % nblobs: a 1M-row x 602 KB table is 602 GB, so the
// rows cycle over a table that fits) and run `SELECT infera_predict_from_blob(model, img)` on each through infera_sql_call --
// one batched engine call per chunk, pipelined pinned staging, H2D, the conv net, D2H, LIST result.  0 / -1.
int32_t infera_sql_bench_blob_scan(const char *model, const uint8_t *blobs, uint64_t nblobs, uint64_t blob_bytes, uint64_t rows,
                                   int32_t threads, int32_t reps, double *secs, double *checksum, char *err, uint64_t errlen) {
  if (threads < 1) threads = 1;
  const size_t CH = INFERA_SQL_VECTOR_SIZE;
  const uint64_t nchunks = (rows + CH - 1) / CH;
  std::string first_error;
  std::mutex mu;
  for (int rep = 0; rep < reps; rep++) {
    std::atomic<uint64_t> next{0};
    double total = 0.0;
    auto worker = [&] {
      InferaSqlVector args[2];
      const uint8_t *name_ptr = reinterpret_cast<const uint8_t *>(model);
      uint64_t name_len = std::strlen(model);
      args[0] = InferaSqlVector{INFERA_SQL_VARCHAR, 1, &name_ptr, &name_len, nullptr};
      std::vector<const uint8_t *> ptrs(CH);
      std::vector<uint64_t> lens(CH, blob_bytes);
      double local = 0.0;
      for (;;) {
        const uint64_t c = next.fetch_add(1, std::memory_order_relaxed);
        if (c >= nchunks) break;
        const size_t nr = size_t(std::min<uint64_t>(CH, rows - c * CH));
        for (size_t i = 0; i < nr; i++) ptrs[i] = blobs + ((c * CH + i) % nblobs) * blob_bytes;
        args[1] = InferaSqlVector{INFERA_SQL_BLOB, 0, ptrs.data(), lens.data(), nullptr};
        InferaSqlResult res;
        if (infera_sql_call("infera_predict_from_blob", args, 2, nr, &res) != 0) {
          std::lock_guard<std::mutex> lk(mu);
          if (first_error.empty()) first_error = res.error ? res.error : "unknown error";
          infera_sql_free_result(&res);
          next.store(nchunks);
          break;
        }
        if (res.list_offsets) local += sum_block(res.list_values, size_t(res.list_offsets[nr]));
        infera_sql_free_result(&res);
      }
      std::lock_guard<std::mutex> lk(mu);
      total += local;
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker);
    for (auto &x : th) x.join();
    if (secs) secs[rep] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (checksum) *checksum = total;
    if (!first_error.empty()) {
      if (err && errlen) std::snprintf(err, size_t(errlen), "%s", first_error.c_str());
      return -1;
    }
  }
  return 0;
}


}  // extern "C"
