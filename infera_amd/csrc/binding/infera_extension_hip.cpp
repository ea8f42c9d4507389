// infera_extension_hip.cpp -- the DuckDB extension of the MI355X backend: the SQL surface of Infera over DuckDB's
// vector API, on top of the C ABI of include/infera.h + include/infera_hip.h (libinfera.so).
//
// Replaces /root/reference infera/bindings/infera_extension.cpp as a whole.  Same 13 SQL functions,
// same argument checks, NULL behaviour and error strings (the reference's sqllogictests pin them; each function below
// cites the lines it answers to), plus `infera_predict_array` and feature overloads up to INFERA_MAX_FEATURES.
// What is different is how a DataChunk crosses the boundary:
//
//   * features: the reference boxes every cell (`Vector::GetValue`, 262,144 `duckdb::Value`s per 2048 x 128 chunk,
//     infera_extension.cpp:199-227).  Here every argument vector goes through `ToUnifiedFormat`: a FLAT vector is handed
//     to infera_predict_columns as the pointer it already is (zero gather work in the binding; the engine copies each
//     8 KiB column run into pinned staging and the GPU transposes), a CONSTANT vector as one element, a dictionary /
//     sliced vector is compacted through its selection vector first; validity masks are tested a word at a time;
//     DECIMAL columns are cast once per vector (`VectorOperations::DefaultCast`), not once per cell.
//   * LIST results: children are written in one block (`ListVector::Reserve` + one memcpy + `list_entry_t` per row)
//     instead of `Value::FLOAT` per element + `Value::LIST` + `SetValue` per row (infera_extension.cpp:451-459).
//   * BLOBs: one `infera_predict_from_blob_batch` call per chunk when the chunk uses one model and every blob holds one
//     sample, instead of one FFI call and one batch-1 run per row (infera_extension.cpp:303-326).
//
// Built only when DuckDB's headers are present (binding/CMakeLists.txt); the repository's tests compile and drive this
// very file against tests/duckdb_stub/ (a minimal stand-in for the handful of DuckDB types used here -- test
// infrastructure, never shipped).
#define DUCKDB_EXTENSION_MAIN

#include "duckdb.hpp"
#include "duckdb/common/allocator.hpp"
#include "duckdb/common/exception.hpp"
#include "duckdb/common/types/data_chunk.hpp"
#include "duckdb/common/types/vector.hpp"
#include "duckdb/common/vector_operations/vector_operations.hpp"
#include "duckdb/function/scalar_function.hpp"
#include "duckdb/main/config.hpp"
#include "duckdb/main/extension/extension_loader.hpp"

#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "infera_hip.h"  // namespace infera: the 13 reference symbols + the additive entry points

#ifndef INFERA_MAX_FEATURES
#define INFERA_MAX_FEATURES 256  // reference: 127 (infera_extension.cpp:550) -- BASELINE config C2 has 128 columns
#endif

namespace duckdb {

namespace {

std::string LastError() {  // infera_extension.cpp:52-55
  const char *err = infera::infera_last_error();
  return err ? std::string(err) : std::string("unknown error");
}

// RAII for the two ownership rules of the boundary (rust.h:299-316)
struct ResultGuard {
  infera::InferaInferenceResult res;
  explicit ResultGuard(infera::InferaInferenceResult r) : res(r) {}
  ~ResultGuard() { infera::infera_free_result(res); }  // callers free even failed results (NULL data is a no-op)
  ResultGuard(const ResultGuard &) = delete;
  ResultGuard &operator=(const ResultGuard &) = delete;
};
std::string TakeString(char *s, const char *if_null = "") {
  std::string out = s ? std::string(s) : std::string(if_null);
  infera::infera_free(s);
  return out;
}

// ---- reading arguments ------------------------------------------------------------------------------------------
// Row `row` of a VARCHAR / BLOB argument vector whatever its physical form.
struct StringArg {
  UnifiedVectorFormat fmt;
  StringArg(Vector &v, idx_t count) { v.ToUnifiedFormat(count, fmt); }
  bool IsNull(idx_t row) const { return !fmt.validity.RowIsValid(fmt.sel->get_index(row)); }
  string_t Get(idx_t row) const { return UnifiedVectorFormat::GetData<string_t>(fmt)[fmt.sel->get_index(row)]; }
  std::string GetString(idx_t row) const {
    const string_t s = Get(row);
    return std::string(s.GetData(), s.GetSize());
  }
};

// "requires a non-NULL value in row 0" -- the check every management function makes (infera_extension.cpp:94-97, ...)
std::string RequireString(DataChunk &args, idx_t col, const char *null_message) {
  StringArg a(args.data[col], args.size());
  if (a.IsNull(0)) throw InvalidInputException(null_message);
  return a.GetString(0);
}

void SetConstantString(Vector &result, const std::string &s) {
  result.SetVectorType(VectorType::CONSTANT_VECTOR);
  ConstantVector::GetData<string_t>(result)[0] = StringVector::AddString(result, s);
  ConstantVector::SetNull(result, false);
}
void SetConstantBool(Vector &result, bool v) {
  result.SetVectorType(VectorType::CONSTANT_VECTOR);
  ConstantVector::GetData<bool>(result)[0] = v;
  ConstantVector::SetNull(result, false);
}

// ---- ExtractFeatures without boxing (answers infera_extension.cpp:199-227) -----------------------------------------
// One feature column of the chunk as the engine's InferaColumn.  `keep` owns whatever had to be materialised (a DECIMAL
// column cast to DOUBLE, a dictionary vector compacted through its selection vector).
struct FeatureColumns {
  std::vector<infera::InferaColumn> cols;
  std::unique_ptr<UnifiedVectorFormat[]> fmt;  // (not a std::vector: UnifiedVectorFormat is not copyable)
  std::vector<unique_ptr<Vector>> casts;
  std::vector<std::vector<uint8_t>> compacted;

  static bool AnyNull(const UnifiedVectorFormat &f, idx_t count, bool flat) {
    static_assert(sizeof(validity_t) == 8, "the word-at-a-time NULL scan assumes DuckDB's 64-bit validity words");
    if (f.validity.AllValid()) return false;
    if (flat) {  // whole validity words (DuckDB: bit set = valid, 64 rows per word -- the engine's format too)
      const auto *w = f.validity.GetData();
      const idx_t full = count / 64, rest = count % 64;
      for (idx_t i = 0; i < full; i++)
        if (w[i] != ~validity_t(0)) return true;
      const validity_t mask = rest ? ((validity_t(1) << rest) - 1) : 0;
      return rest && (w[full] & mask) != mask;
    }
    for (idx_t r = 0; r < count; r++)
      if (!f.validity.RowIsValid(f.sel->get_index(r))) return true;
    return false;
  }

  FeatureColumns(DataChunk &args, idx_t count) {
    const idx_t F = args.ColumnCount() - 1;
    cols.resize(F);
    fmt.reset(new UnifiedVectorFormat[F]);
    for (idx_t c = 0; c < F; c++) {
      Vector *v = &args.data[c + 1];
      int32_t type;
      size_t width;
      switch (v->GetType().id()) {
        case LogicalTypeId::FLOAT: type = infera::INFERA_COL_FLOAT; width = 4; break;
        case LogicalTypeId::DOUBLE: type = infera::INFERA_COL_DOUBLE; width = 8; break;
        case LogicalTypeId::INTEGER: type = infera::INFERA_COL_INTEGER; width = 4; break;
        case LogicalTypeId::BIGINT: type = infera::INFERA_COL_BIGINT; width = 8; break;
        case LogicalTypeId::DECIMAL: {  // :214-219 -- DECIMAL -> DOUBLE -> float, once per vector
          casts.push_back(make_uniq<Vector>(LogicalType::DOUBLE, count));
          VectorOperations::DefaultCast(*v, *casts.back(), count);
          v = casts.back().get();
          type = infera::INFERA_COL_DOUBLE;
          width = 8;
          break;
        }
        default: throw InvalidInputException("Unsupported feature type: " + v->GetType().ToString());  // :221-222
      }
      const bool constant = v->GetVectorType() == VectorType::CONSTANT_VECTOR;
      UnifiedVectorFormat &f = fmt[c];
      v->ToUnifiedFormat(count, f);
      const bool flat = !constant && !f.sel->IsSet();  // incremental selection = the buffer is the column
      if (constant ? !f.validity.RowIsValid(f.sel->get_index(0)) : AnyNull(f, count, flat))
        throw InvalidInputException("Feature values cannot be NULL");  // :207-209
      infera::InferaColumn &col = cols[c];
      col.type = type;
      col.validity = nullptr;  // checked above, with the reference's error text and precedence
      col.is_constant = constant ? 1 : 0;
      if (constant) {
        col.data = f.data + size_t(f.sel->get_index(0)) * width;
      } else if (flat) {
        col.data = f.data;  // zero-copy: the engine reads DuckDB's own buffer for the duration of the call
      } else {              // dictionary / sliced vector: compact through the selection vector
        compacted.emplace_back(size_t(count) * width);
        uint8_t *dst = compacted.back().data();
        if (width == 4) {
          auto *d = reinterpret_cast<uint32_t *>(dst);
          const auto *s = reinterpret_cast<const uint32_t *>(f.data);
          for (idx_t r = 0; r < count; r++) d[r] = s[f.sel->get_index(r)];
        } else {
          auto *d = reinterpret_cast<uint64_t *>(dst);
          const auto *s = reinterpret_cast<const uint64_t *>(f.data);
          for (idx_t r = 0; r < count; r++) d[r] = s[f.sel->get_index(r)];
        }
        col.data = dst;
      }
    }
  }
};

// ValidateAndGetModelName (infera_extension.cpp:239-248): >= 2 columns, name from ROW 0 only, non-NULL
std::string ValidateAndGetModelName(DataChunk &args, const std::string &func_name) {
  if (args.ColumnCount() < 2) throw InvalidInputException(func_name + "(model_name, feature1, ...) requires at least 2 arguments");
  return RequireString(args, 0, "Model name cannot be NULL");
}

// gather + one FFI call for the whole chunk; throws with the reference's text on failure (:270-274)
infera::InferaInferenceResult RunChunk(DataChunk &args, const std::string &model) {
  const idx_t rows = args.size();
  FeatureColumns features(args, rows);
  infera::InferaInferenceResult res = infera::infera_predict_columns(model.c_str(), features.cols.data(), features.cols.size(), rows);
  if (res.status != 0) {
    infera::infera_free_result(res);
    throw InvalidInputException("Inference failed for model '" + model + "': " + LastError());
  }
  return res;
}

std::string FormatCounts(const char *fmt, unsigned long long a, unsigned long long b, unsigned long long c = 0) {
  char buf[192];
  std::snprintf(buf, sizeof buf, fmt, a, b, c);  // StringUtil::Format("%d") of idx_t / size_t values (:276, :398, :446)
  return buf;
}

// ---- the predict family -------------------------------------------------------------------------------------------

// infera_predict(model, f1..fN) -> FLOAT   (answers infera_extension.cpp:260-286)
void Predict(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.size() == 0) return;
  const std::string model = ValidateAndGetModelName(args, "infera_predict");
  const idx_t rows = args.size();
  ResultGuard g(RunChunk(args, model));
  if (g.res.rows != rows || g.res.cols != 1)
    throw InvalidInputException(FormatCounts("Model output shape mismatch. Expected (%llu, 1), but got (%llu, %llu).", rows, g.res.rows, g.res.cols));
  result.SetVectorType(VectorType::FLAT_VECTOR);
  std::memcpy(FlatVector::GetData<float>(result), g.res.data, size_t(rows) * sizeof(float));
}

// infera_predict_multi(model, f1..fN) -> VARCHAR "[a,b,...]"   (answers :382-418; `ostream << float` == %g, 6 digits)
void PredictMulti(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.size() == 0) return;
  const std::string model = ValidateAndGetModelName(args, "infera_predict_multi");
  const idx_t rows = args.size();
  ResultGuard g(RunChunk(args, model));
  if (g.res.rows != rows) throw InvalidInputException(FormatCounts("Model output row count mismatch. Expected %llu, but got %llu.", rows, g.res.rows));
  result.SetVectorType(VectorType::FLAT_VECTOR);
  auto *out = FlatVector::GetData<string_t>(result);
  const size_t cols = g.res.cols;
  std::vector<char> line(cols * 17 + 3);
  for (idx_t r = 0; r < rows; r++) {
    char *p = line.data();
    *p++ = '[';
    for (size_t c = 0; c < cols; c++) {
      if (c) *p++ = ',';
      p = std::to_chars(p, p + 16, g.res.data[size_t(r) * cols + c], std::chars_format::general, 6).ptr;
    }
    *p++ = ']';
    out[r] = StringVector::AddString(result, line.data(), idx_t(p - line.data()));
  }
}

// rows x cols floats -> LIST<FLOAT>[rows], children written as one block
void WriteListBlock(Vector &result, const float *data, idx_t rows, idx_t cols) {
  result.SetVectorType(VectorType::FLAT_VECTOR);
  const idx_t total = rows * cols;
  ListVector::Reserve(result, total);
  if (total) std::memcpy(FlatVector::GetData<float>(ListVector::GetEntry(result)), data, size_t(total) * sizeof(float));
  auto *entries = FlatVector::GetData<list_entry_t>(result);
  for (idx_t r = 0; r < rows; r++) entries[r] = list_entry_t(r * cols, cols);
  ListVector::SetListSize(result, total);
}

// infera_predict_multi_list / infera_predict_array (model, f1..fN) -> FLOAT[]   (answers :430-462)
template <const char *NAME>
void PredictMultiList(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.size() == 0) return;
  const std::string model = ValidateAndGetModelName(args, NAME);
  const idx_t rows = args.size();
  ResultGuard g(RunChunk(args, model));
  if (g.res.rows != rows) throw InvalidInputException(FormatCounts("Model output row count mismatch. Expected %llu, but got %llu.", rows, g.res.rows));
  WriteListBlock(result, g.res.data, rows, g.res.cols);
}
constexpr char kMultiListName[] = "infera_predict_multi_list";
constexpr char kArrayName[] = "infera_predict_array";

// infera_predict_from_blob(model, blob) -> FLOAT[]   (answers :297-328; NULL name or blob -> NULL result row)
void PredictFromBlob(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.ColumnCount() != 2) throw InvalidInputException("infera_predict_from_blob(model_name, input_blob) requires 2 arguments");
  const idx_t rows = args.size();
  if (rows == 0) return;
  StringArg names(args.data[0], rows), blobs(args.data[1], rows);
  std::vector<idx_t> live;
  live.reserve(rows);
  for (idx_t r = 0; r < rows; r++)
    if (!names.IsNull(r) && !blobs.IsNull(r)) live.push_back(r);
  result.SetVectorType(VectorType::FLAT_VECTOR);
  auto *entries = FlatVector::GetData<list_entry_t>(result);
  auto &validity = FlatVector::Validity(result);
  for (idx_t r = 0; r < rows; r++) entries[r] = list_entry_t(0, 0);
  {
    idx_t k = 0;
    for (idx_t r = 0; r < rows; r++) {
      if (k < live.size() && live[k] == r) k++;
      else validity.SetInvalid(r);
    }
  }
  if (live.empty()) {
    ListVector::SetListSize(result, 0);
    return;
  }
  // One FFI call and one GPU batch for the chunk when it uses one model and every blob is one sample
  bool one_model = true;
  const string_t first = names.Get(live[0]);
  for (idx_t r : live) {
    const string_t n = names.Get(r);
    one_model = one_model && n.GetSize() == first.GetSize() && std::memcmp(n.GetData(), first.GetData(), first.GetSize()) == 0;
  }
  if (one_model && live.size() > 1) {
    const std::string model = names.GetString(live[0]);
    std::vector<const uint8_t *> ptrs(live.size());
    std::vector<uintptr_t> lens(live.size());
    for (size_t k = 0; k < live.size(); k++) {
      const string_t b = blobs.Get(live[k]);
      ptrs[k] = reinterpret_cast<const uint8_t *>(b.GetData());
      lens[k] = b.GetSize();
    }
    ResultGuard g(infera::infera_predict_from_blob_batch(model.c_str(), ptrs.data(), lens.data(), live.size()));
    if (g.res.status == 0 && g.res.rows == live.size()) {
      const idx_t cols = g.res.cols, total = idx_t(live.size()) * cols;
      ListVector::Reserve(result, total);
      if (total) std::memcpy(FlatVector::GetData<float>(ListVector::GetEntry(result)), g.res.data, size_t(total) * sizeof(float));
      for (size_t k = 0; k < live.size(); k++) entries[live[k]] = list_entry_t(k * cols, cols);
      ListVector::SetListSize(result, total);
      return;
    }
    // not one sample per blob (or a per-row error): the reference's per-row route below also reproduces its messages
  }
  idx_t total = 0;
  for (idx_t r : live) {
    const std::string model = names.GetString(r);
    const string_t b = blobs.Get(r);
    ResultGuard g(infera::infera_predict_from_blob(model.c_str(), reinterpret_cast<const uint8_t *>(b.GetData()), b.GetSize()));
    if (g.res.status != 0) throw InvalidInputException("Inference failed for model '" + model + "': " + LastError());  // :315-318
    ListVector::Reserve(result, total + g.res.len);  // all res.len elements of the row's output (:319-325)
    if (g.res.len) std::memcpy(FlatVector::GetData<float>(ListVector::GetEntry(result)) + total, g.res.data, g.res.len * sizeof(float));
    entries[r] = list_entry_t(total, g.res.len);
    total += g.res.len;
  }
  ListVector::SetListSize(result, total);
}

// ---- model management (answers infera_extension.cpp:87-121, :133-188, :339-380, :473-536) -----------------------------

void LoadModel(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.ColumnCount() != 2) throw InvalidInputException("infera_load_model(model_name, path) expects exactly 2 arguments");
  if (args.size() == 0) return;
  StringArg name(args.data[0], args.size()), path(args.data[1], args.size());
  if (name.IsNull(0) || path.IsNull(0)) throw InvalidInputException("Model name and path cannot be NULL");
  const std::string model = name.GetString(0), file = path.GetString(0);
  if (model.empty()) throw InvalidInputException("Model name cannot be empty");
  if (infera::infera_load_model(model.c_str(), file.c_str()) != 0)
    throw InvalidInputException("Failed to load model '" + model + "': " + LastError());
  SetConstantBool(result, true);
}

void UnloadModel(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.ColumnCount() != 1) throw InvalidInputException("infera_unload_model(model_name) expects exactly 1 argument");
  if (args.size() == 0) return;
  const std::string model = RequireString(args, 0, "Model name cannot be NULL");
  if (infera::infera_unload_model(model.c_str()) != 0) {
    const std::string err = LastError();
    if (err.rfind("Model not found:", 0) != 0)  // unloading a missing model is a benign no-op (:179-183)
      throw InvalidInputException("Failed to unload model '" + model + "': " + err);
  }
  SetConstantBool(result, true);
}

void GetModelInfo(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.ColumnCount() != 1) throw InvalidInputException("infera_get_model_info(model_name) expects exactly 1 argument");
  if (args.size() == 0) return;
  const std::string model = RequireString(args, 0, "Model name cannot be NULL");
  const std::string json = TakeString(infera::infera_get_model_info(model.c_str()));
  if (json.empty() || json.find("\"error\"") != std::string::npos)  // error JSON -> SQL error (:491-494)
    throw InvalidInputException("Failed to get info for model '" + model + "'");
  SetConstantString(result, json);
}

void IsModelLoaded(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.ColumnCount() != 1) throw InvalidInputException("infera_is_model_loaded(model_name) expects exactly 1 argument");
  if (args.size() == 0) return;
  const std::string model = RequireString(args, 0, "Model name cannot be NULL");
  const std::string models = TakeString(infera::infera_get_loaded_models());
  SetConstantBool(result, models.find("\"" + model + "\"") != std::string::npos);  // quoted-name search (:373-374)
}

void SetAutoloadDir(DataChunk &args, ExpressionState &, Vector &result) {
  if (args.ColumnCount() != 1) throw InvalidInputException("infera_set_autoload_dir(path) expects exactly 1 argument");
  if (args.size() == 0) return;
  const std::string path = RequireString(args, 0, "Path cannot be NULL");
  SetConstantString(result, TakeString(infera::infera_set_autoload_dir(path.c_str())));
}

void GetLoadedModels(DataChunk &, ExpressionState &, Vector &result) { SetConstantString(result, TakeString(infera::infera_get_loaded_models(), "[]")); }
void GetVersion(DataChunk &, ExpressionState &, Vector &result) { SetConstantString(result, TakeString(infera::infera_get_version())); }
void GetCacheInfo(DataChunk &, ExpressionState &, Vector &result) { SetConstantString(result, TakeString(infera::infera_get_cache_info())); }
void ClearCache(DataChunk &, ExpressionState &, Vector &result) {
  if (infera::infera_clear_cache() != 0) throw InvalidInputException("Failed to clear cache: " + LastError());
  SetConstantBool(result, true);
}

// ---- registration (answers infera_extension.cpp:64-75, :546-592) -----------------------------------------------------

// Volatile + fallible metadata across DuckDB versions: newer trees have SetVolatile()/SetFallible(), older ones the
// `stability` / `errors` members.
template <int N> struct Rank : Rank<N - 1> {};
template <> struct Rank<0> {};
template <class F> auto MarkVolatile(F &f, Rank<1>) -> decltype(f.SetVolatile(), void()) { f.SetVolatile(); }
template <class F> auto MarkVolatile(F &f, Rank<0>) -> decltype(f.stability = FunctionStability::VOLATILE, void()) { f.stability = FunctionStability::VOLATILE; }
template <class F> auto MarkFallible(F &f, Rank<1>) -> decltype(f.SetFallible(), void()) { f.SetFallible(); }
template <class F> auto MarkFallible(F &f, Rank<0>) -> decltype(f.errors = FunctionErrors::CAN_THROW_RUNTIME_ERROR, void()) { f.errors = FunctionErrors::CAN_THROW_RUNTIME_ERROR; }

ScalarFunction Make(const std::string &name, vector<LogicalType> arguments, LogicalType return_type, scalar_function_t fn,
                    bool volatile_state, bool fallible) {
  ScalarFunction f(name, std::move(arguments), std::move(return_type), std::move(fn));
  if (volatile_state) MarkVolatile(f, Rank<1>{});
  if (fallible) MarkFallible(f, Rank<1>{});
  return f;
}

void LoadInternal(ExtensionLoader &loader) {
  loader.RegisterFunction(Make("infera_load_model", {LogicalType::VARCHAR, LogicalType::VARCHAR}, LogicalType::BOOLEAN, LoadModel, true, true));
  loader.RegisterFunction(Make("infera_unload_model", {LogicalType::VARCHAR}, LogicalType::BOOLEAN, UnloadModel, true, true));

  // The predict family: one overload per feature count and per {FLOAT, DOUBLE} (the DOUBLE set gives DECIMAL literals a
  // bind path, infera_extension.cpp:564-577), all volatile (they read the live model registry) and fallible.
  struct Family {
    const char *name;
    LogicalType returns;
    scalar_function_t fn;
  };
  const LogicalType list_of_float = LogicalType::LIST(LogicalType::FLOAT);
  const Family families[] = {{"infera_predict", LogicalType::FLOAT, Predict},
                             {"infera_predict_multi", LogicalType::VARCHAR, PredictMulti},
                             {"infera_predict_multi_list", list_of_float, PredictMultiList<kMultiListName>},
                             {"infera_predict_array", list_of_float, PredictMultiList<kArrayName>}};
  for (const Family &fam : families) {
    ScalarFunctionSet set(fam.name);
    for (idx_t n = 1; n <= INFERA_MAX_FEATURES; n++)
      for (const LogicalTypeId feature_type : {LogicalType::FLOAT, LogicalType::DOUBLE}) {
        vector<LogicalType> types;
        types.reserve(n + 1);
        types.push_back(LogicalType::VARCHAR);
        for (idx_t i = 0; i < n; i++) types.push_back(LogicalType(feature_type));
        set.AddFunction(Make(fam.name, std::move(types), fam.returns, fam.fn, true, true));
      }
    loader.RegisterFunction(set);
  }

  loader.RegisterFunction(Make("infera_predict_from_blob", {LogicalType::VARCHAR, LogicalType::BLOB}, list_of_float, PredictFromBlob, true, true));
  loader.RegisterFunction(Make("infera_get_loaded_models", {}, LogicalType::VARCHAR, GetLoadedModels, true, false));
  loader.RegisterFunction(Make("infera_get_model_info", {LogicalType::VARCHAR}, LogicalType::VARCHAR, GetModelInfo, true, true));
  loader.RegisterFunction(Make("infera_get_version", {}, LogicalType::VARCHAR, GetVersion, false, false));
  loader.RegisterFunction(Make("infera_set_autoload_dir", {LogicalType::VARCHAR}, LogicalType::VARCHAR, SetAutoloadDir, true, true));
  loader.RegisterFunction(Make("infera_is_model_loaded", {LogicalType::VARCHAR}, LogicalType::BOOLEAN, IsModelLoaded, true, false));
  loader.RegisterFunction(Make("infera_clear_cache", {}, LogicalType::BOOLEAN, ClearCache, true, true));
  loader.RegisterFunction(Make("infera_get_cache_info", {}, LogicalType::VARCHAR, GetCacheInfo, true, false));
}

}  // namespace

// ---- zero-copy allocator (opt-in; the reference's ROADMAP.md:44 "zero-copy") ------------------------------------------------------
// DuckDB takes every buffer-manager block -- the memory column segments of a scan live in -- from DBConfig::allocator.  An allocator that
// REGISTERS the blocks it hands out (infera_hip_register_host_memory: pinned where they lie, mapped into every GPU, nothing copied) makes a
// chunk whose column vectors all point into such blocks readable in place by the GPU: infera_predict_columns then costs the CPU no gather
// and no H2D enqueue (16-28 us per 2048 x 128 chunk instead of 70).  Registration never stalls running scans and freeing a
// block waits only for the calls that are reading THAT block (hip/zero_copy.cpp registry; tests/native/scan_stress.cpp).  Blocks smaller than
// `min_bytes` (default 128 KiB: a column segment is 256 KiB) are not worth a registration and stay ordinary memory.
// The allocator has to be in place BEFORE the database is opened -- an extension cannot swap it afterwards -- so this is for the embedding
// application:   DBConfig config;  infera_install_zero_copy_allocator(config);  DuckDB db(path, &config);
// It is a no-op unless INFERA_ZERO_COPY_ALLOCATOR=1 (a registered block stays pinned for its lifetime: opt-in, like the path itself).
namespace {
// ARENA (round 6).  Registering every 256 KiB block by itself costs an ioctl per Allocate / Free (~28 us each, on whichever thread DuckDB's
// buffer manager allocates), leaves 20,000 registrations for a 5 GB table, and scatters a chunk's 128 column vectors over 128 unrelated
// registrations -- which only the pulling kernel can fetch.  So blocks of the buffer manager's size (`block_bytes`, default 256 KiB = DuckDB's
// Storage::BLOCK_ALLOC_SIZE) come out of SLABS of 256 blocks (64 MiB), each registered ONCE: Allocate / Free are a bit operation under a mutex,
// the registry holds one range per slab, and blocks handed out back to back -- the column segments of a row group loaded by one thread -- lie at
// ONE stride inside ONE registration, which is what the engine's 2-D copy needs (include/infera_hip.h: FLOAT runs at one pitch inside one pinned
// block).  Lowest free block of the oldest slab with room first, so that consecutive allocations stay adjacent and emptied slabs can go back to
// the system (one empty slab is kept as hysteresis).  Other sizes >= `min_bytes` keep their own registration; smaller ones are plain memory.
// A slab's pages are pinned as a whole: up to 64 MiB of slack per arena.  INFERA_ZERO_COPY_ARENA=0: the per-block registrations of round 4.
struct Slab {
  data_ptr_t base = nullptr;
  uint64_t free_bits[4] = {~uint64_t(0), ~uint64_t(0), ~uint64_t(0), ~uint64_t(0)};  // bit set = free
  int used = 0;
  bool registered = false;
};
struct InferaAllocatorData : PrivateAllocatorData {
  idx_t min_bytes = 128 * 1024;
  idx_t block_bytes = 256 * 1024;
  bool arena = true;
  static constexpr int kSlabBlocks = 256;
  std::mutex mu;
  std::vector<unique_ptr<Slab>> slabs;      // creation order
  std::map<data_ptr_t, Slab *> by_base;     // (a 100 GB buffer pool is 1,600 slabs: Free finds its slab in O(log n))
  ~InferaAllocatorData() override {
    for (auto &sl : slabs) {
      if (sl->registered) (void)infera::infera_hip_unregister_host_memory(sl->base);
      std::free(sl->base);
    }
  }
  Slab *SlabOf(data_ptr_t p) {  // (under mu)
    auto it = by_base.upper_bound(p);
    if (it == by_base.begin()) return nullptr;
    --it;
    return p < it->first + idx_t(kSlabBlocks) * block_bytes ? it->second : nullptr;
  }
  data_ptr_t Take() {
    {
      std::lock_guard<std::mutex> lk(mu);
      for (auto &sl : slabs)
        if (sl->used < kSlabBlocks)
          for (int w = 0; w < 4; w++)
            if (sl->free_bits[w]) {
              const int bit = __builtin_ctzll(sl->free_bits[w]);
              sl->free_bits[w] &= ~(uint64_t(1) << bit);
              sl->used++;
              return sl->base + idx_t(64 * w + bit) * block_bytes;
            }
    }
    // no room: a new slab, made and pinned OUTSIDE the lock (pinning 64 MiB takes milliseconds; other threads keep allocating and freeing)
    auto fresh = make_uniq<Slab>();
    void *mem = nullptr;
    if (posix_memalign(&mem, size_t(2) << 20, size_t(kSlabBlocks) * block_bytes) != 0) return nullptr;
    fresh->base = static_cast<data_ptr_t>(mem);
    // (a failed registration -- no GPU, pinning limit reached -- only means the slab's chunks are staged like any other memory)
    fresh->registered = infera::infera_hip_register_host_memory(mem, uint64_t(kSlabBlocks) * block_bytes) == 0;
    fresh->free_bits[0] &= ~uint64_t(1);
    fresh->used = 1;
    std::lock_guard<std::mutex> lk(mu);
    slabs.push_back(std::move(fresh));
    by_base[slabs.back()->base] = slabs.back().get();
    return slabs.back()->base;
  }
  // true: `p` was a slab block and has been given back
  bool GiveBack(data_ptr_t p) {
    unique_ptr<Slab> retired;
    {
      std::lock_guard<std::mutex> lk(mu);
      Slab *sl = SlabOf(p);
      if (!sl) return false;
      const idx_t i = idx_t(p - sl->base) / block_bytes;
      sl->free_bits[i / 64] |= uint64_t(1) << (i % 64);
      sl->used--;
      if (sl->used == 0) {  // keep ONE empty slab around; a second one goes back to the system
        int empty = 0;
        for (auto &x : slabs) empty += x->used == 0;
        if (empty > 1)
          for (auto it = slabs.begin(); it != slabs.end(); ++it)
            if (it->get() == sl) {
              by_base.erase(sl->base);
              retired = std::move(*it);
              slabs.erase(it);
              break;
            }
      }
    }
    if (retired) {  // outside the lock: unregistering waits for the calls still reading the slab (none: its blocks are all free)
      bool release = true;
      if (retired->registered && infera::infera_hip_unregister_host_memory(retired->base) != 0) {
        const char *e = infera::infera_last_error();
        release = !(e && std::strstr(e, "still in use"));  // (a wedged GPU may still read it: leaked, include/infera_hip.h)
      }
      if (release) std::free(retired->base);
    }
    return true;
  }
};
inline InferaAllocatorData *Data(PrivateAllocatorData *pd) { return static_cast<InferaAllocatorData *>(pd); }
inline bool WorthRegistering(PrivateAllocatorData *pd, idx_t size) { return size >= Data(pd)->min_bytes; }
inline bool FromArena(PrivateAllocatorData *pd, idx_t size) { return Data(pd)->arena && size == Data(pd)->block_bytes; }
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wmaybe-uninitialized"  // (the fresh block's CONTENTS are uninitialised; only its address range is registered)
#endif
data_ptr_t RegisteringAllocate(PrivateAllocatorData *pd, idx_t size) {
  if (FromArena(pd, size)) return Data(pd)->Take();
  const data_ptr_t p = Allocator::DefaultAllocate(pd, size);
  // (a failed registration -- no GPU, pinning limit reached -- only means the block's chunks are staged like any other memory)
  if (p && WorthRegistering(pd, size)) (void)infera::infera_hip_register_host_memory(p, size);
  return p;
}
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC diagnostic pop
#endif
// Unregisters a block before it goes back to the heap (waits for the calls still reading it: a chunk's time).  false: calls were STILL reading it
// after the library's bounded wait (a wedged GPU) -- the memory may not be freed or moved; the block is leaked instead (include/infera_hip.h:
// only a 0 from infera_hip_unregister_host_memory says the range is no longer read).  "Was not registered" (no GPU, pinning limit) is fine.
bool ReleasedForFree(data_ptr_t p) {
  if (infera::infera_hip_unregister_host_memory(p) == 0) return true;
  const char *e = infera::infera_last_error();
  return !(e && std::strstr(e, "still in use"));
}
void RegisteringFree(PrivateAllocatorData *pd, data_ptr_t p, idx_t size) {
  if (!p) return;
  if (FromArena(pd, size) && Data(pd)->GiveBack(p)) return;
  if (WorthRegistering(pd, size) && !ReleasedForFree(p)) return;
  Allocator::DefaultFree(pd, p, size);
}
data_ptr_t RegisteringReallocate(PrivateAllocatorData *pd, data_ptr_t p, idx_t old_size, idx_t size) {
  if (!p) return RegisteringAllocate(pd, size);
  const idx_t keep = old_size < size ? old_size : size;
  if (FromArena(pd, old_size) || FromArena(pd, size)) {
    // a slab block cannot grow in place, and a block of the arena's size has to come out of a slab: allocate, copy, release the old one
    const data_ptr_t fresh = RegisteringAllocate(pd, size);
    if (!fresh) return nullptr;
    std::memcpy(fresh, p, keep);
    RegisteringFree(pd, p, old_size);
    return fresh;
  }
  if (WorthRegistering(pd, old_size) && !ReleasedForFree(p)) {  // a wedged GPU may still read the old block: it stays where it is (leaked)
    const data_ptr_t fresh = RegisteringAllocate(pd, size);
    if (fresh) std::memcpy(fresh, p, keep);
    return fresh;
  }
  const data_ptr_t q = Allocator::DefaultReallocate(pd, p, old_size, size);
  if (q && WorthRegistering(pd, size)) (void)infera::infera_hip_register_host_memory(q, size);
  return q;
}
}  // namespace

// The Extension subclass DuckDB's static-extension loader instantiates (infera/bindings/include/infera_extension.hpp).
class InferaExtension : public Extension {
public:
  void Load(ExtensionLoader &loader) override { LoadInternal(loader); }
  std::string Name() override { return "infera"; }
  std::string Version() const override { return "v0.4.0-hip"; }
};

}  // namespace duckdb

extern "C" {
// Entry point of a C++ loadable extension (infera_extension.cpp:600-610)
DUCKDB_EXTENSION_API void infera_duckdb_cpp_init(duckdb::ExtensionLoader &loader) { duckdb::LoadInternal(loader); }
DUCKDB_EXTENSION_API void infera_init(duckdb::DatabaseInstance &db) {
  duckdb::ExtensionLoader loader(db, "infera");
  duckdb::LoadInternal(loader);
}
// INFERA_ZERO_COPY_ALLOCATOR=1: config.allocator becomes the registering allocator above; returns whether it was installed
DUCKDB_EXTENSION_API bool infera_install_zero_copy_allocator(duckdb::DBConfig &config) {
  const char *e = std::getenv("INFERA_ZERO_COPY_ALLOCATOR");
  if (!e || std::atoi(e) != 1) return false;
  auto data = duckdb::make_uniq<duckdb::InferaAllocatorData>();
  const char *arena = std::getenv("INFERA_ZERO_COPY_ARENA");
  data->arena = !(arena && std::atoi(arena) == 0);
  config.allocator = duckdb::make_uniq<duckdb::Allocator>(duckdb::RegisteringAllocate, duckdb::RegisteringFree, duckdb::RegisteringReallocate, std::move(data));
  return true;
}
}
