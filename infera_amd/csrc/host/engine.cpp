#include "engine.hpp"

#include <mutex>
#include <shared_mutex>
#include <unordered_map>

#include "../hip/backend.hpp"
#include "common.hpp"

namespace infera_hip {

// engine.rs:19-29: [] -> (1,1); [n] -> (n,1); [d0, ...] -> (d0, max(prod(rest), 1))
std::pair<uint64_t, uint64_t> shape_rows_cols(const std::vector<uint64_t> &shape) {
  if (shape.empty()) return {1, 1};
  if (shape.size() == 1) return {shape[0], 1};
  uint64_t c = 1;
  for (size_t i = 1; i < shape.size(); i++) c *= shape[i];
  return {shape[0], c < 1 ? 1 : c};
}

namespace engine {

namespace {
std::shared_mutex g_mu;
std::unordered_map<std::string, std::shared_ptr<const LoadedModel>> g_models;

// Rust `{:?}` of &[i64]: "[3]" / "[3, 224, 224]"  (engine.rs:132)
std::string debug_i64(const std::vector<int64_t> &v, size_t from) {
  std::string o = "[";
  for (size_t i = from; i < v.size(); i++) o += (i > from ? ", " : "") + std::to_string(v[i]);
  return o + "]";
}
}  // namespace

void load_model(const std::string &name, const std::string &path, const std::string &output_select) {
  // Lowering and the HBM upload happen outside the registry lock; only the insert is exclusive
  // (the reference holds the write lock just for the insert too, engine.rs:80).
  std::shared_ptr<const LoadedModel> m = build_model(name, path, output_select);
  std::unique_lock<std::shared_mutex> lk(g_mu);
  g_models[name] = std::move(m);  // same name silently replaces (engine.rs:74-80)
}

bool unload_model(const std::string &name) {
  std::shared_ptr<const LoadedModel> victim;
  {
    std::unique_lock<std::shared_mutex> lk(g_mu);
    auto it = g_models.find(name);
    if (it == g_models.end()) return false;
    victim = std::move(it->second);
    g_models.erase(it);
  }
  // `victim` (and its HBM) is released here, after in-flight inferences holding a reference finish.
  return true;
}

std::shared_ptr<const LoadedModel> find(const std::string &name) {
  std::shared_lock<std::shared_mutex> lk(g_mu);
  auto it = g_models.find(name);
  if (it == g_models.end()) throw InferaError::model_not_found(name);
  return it->second;
}

std::vector<std::string> loaded_names() {
  std::shared_lock<std::shared_mutex> lk(g_mu);
  std::vector<std::string> out;
  out.reserve(g_models.size());
  for (const auto &kv : g_models) out.push_back(kv.first);
  return out;
}

std::string model_metadata_json(const std::string &name) {
  auto m = find(name);
  // serde_json `json!` object -> keys in sorted order, compact (engine.rs:298-304).  One ADDITIVE key, only for models whose served
  // output the graph declares as an integer tensor (ArgMax labels, Cast to int): the C ABI carries f32 only (rust.h:28-49; the
  // reference rejects such outputs, engine.rs:150-152), so they are served as f32 VALUES -- and the metadata says so.
  std::string extra;
  if (!m->plan.output_declared_type.empty())
    extra = ",\"output_served_as\":" + json_str("f32 values of the graph's " + m->plan.output_declared_type + " output '" + m->plan.output_name + "'");
  return "{\"input_shape\":" + json_int_array(m->plan.input_shape) + ",\"loaded\":true,\"name\":" + json_str(m->name) +
         ",\"output_shape\":" + json_int_array(m->plan.output_shape) + extra + "}";
}

OutShape out_shape_for_rows(const LoadedModel &m, uint64_t rows) {
  std::vector<uint64_t> shp;
  for (size_t i = 0; i < m.plan.output_shape.size(); i++) {
    int64_t d = m.plan.output_shape[i];
    // the leading (row) axis follows the input rows: symbolic, or a fixed batch under INFERA_BATCH_SPLIT
    const bool row_axis = i == 0 && (d < 0 || (m.plan.fixed_batch > 0 && Config::get().batch_split));
    shp.push_back(row_axis || d < 0 ? rows : uint64_t(d));
  }
  OutShape o;
  auto rc = shape_rows_cols(shp);
  o.rows = rc.first;
  o.cols = rc.second;
  o.len = 1;
  for (auto d : shp) o.len *= d;
  return o;
}

OutShape validate_predict(const LoadedModel &m, uint64_t rows, uint64_t cols) {
  const auto &in = m.plan.input_shape;
  // engine.rs:126-137 -- inner dims all known: cols must equal their product
  if (!in.empty()) {
    bool all_known = true;
    uint64_t expected = 1;
    for (size_t i = 1; i < in.size(); i++) {
      if (in[i] <= 0) all_known = false;
      else expected *= uint64_t(in[i]);
    }
    if (all_known && cols != expected)
      throw InferaError::invalid_input_shape("batch x " + debug_i64(in, 1), std::to_string(rows) + " x " + std::to_string(cols));
  }
  // engine.rs:139-145 -- the tensor handed to the backend is always rank-2 [rows, cols]; a model
  // whose input fact has another rank or a different fixed batch rejects it (Tract does this
  // inside run(); the text after "ONNX error: " is this backend's own).
  if (in.size() != 2)
    throw InferaError::onnx("input rank mismatch: model expects rank " + std::to_string(in.size()) + ", got rank 2");
  if (in[0] > 0 && uint64_t(in[0]) != rows && !(Config::get().batch_split && rows % uint64_t(in[0]) == 0))
    throw InferaError::onnx("input shape mismatch at axis 0: model expects " + std::to_string(in[0]) + ", got " + std::to_string(rows));
  if (in[1] > 0 && uint64_t(in[1]) != cols)
    throw InferaError::onnx("input shape mismatch at axis 1: model expects " + std::to_string(in[1]) + ", got " + std::to_string(cols));
  return out_shape_for_rows(m, rows);
}

OutShape validate_device(const LoadedModel &m, uint64_t rows, uint64_t cols) {
  const auto &in = m.plan.input_shape;
  if (in.size() == 2) return validate_predict(m, rows, cols);
  uint64_t per_sample = 1;
  for (size_t i = 1; i < in.size(); i++) per_sample *= uint64_t(in[i]);
  if (cols != per_sample)
    throw InferaError::invalid_input_shape("batch x " + debug_i64(in, 1), std::to_string(rows) + " x " + std::to_string(cols));
  if (!in.empty() && in[0] > 0 && uint64_t(in[0]) != rows)
    throw InferaError::onnx("input shape mismatch at axis 0: model expects " + std::to_string(in[0]) + ", got " + std::to_string(rows));
  return out_shape_for_rows(m, rows);
}

uint64_t validate_blob(const LoadedModel &m, uint64_t blob_len) {
  if (blob_len % 4 != 0) throw InferaError::invalid_blob_size();  // engine.rs:209-211
  const uint64_t n = blob_len / 4;
  const auto &in = m.plan.input_shape;
  uint64_t expected = 1;  // engine.rs:221-226: product of the dims > 0
  for (auto d : in)
    if (d > 0) expected *= uint64_t(d);
  if (expected == 0 || n % expected != 0) throw InferaError::blob_shape_mismatch(size_t(expected), size_t(n));  // :227-232
  const uint64_t batch = n / expected;  // :233
  uint64_t prod = 1;                    // :234-238 every -1 -> batch; Tensor::from_shape must then match
  for (auto d : in) prod *= d == -1 ? batch : uint64_t(d);
  if (prod != n)
    throw InferaError::onnx("shape/data length mismatch: shape holds " + std::to_string(prod) + " elements, data holds " + std::to_string(n));
  // rows for the executor = leading dim after substitution
  return in.empty() ? 1 : (in[0] == -1 ? batch : uint64_t(in[0]));
}

}  // namespace engine
}  // namespace infera_hip
