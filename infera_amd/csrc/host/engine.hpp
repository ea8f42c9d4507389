// engine.hpp -- model registry + inference entry points behind the C ABI.
//
// Mirrors the control flow of the reference's engine.rs / model.rs (not its code):
//   model.rs:41-42      MODELS: global name -> model map, many readers / exclusive writer
//   engine.rs:19-29     shape_rows_cols
//   engine.rs:47-82     load_model_impl    (here: parse -> lower -> upload to HBM)
//   engine.rs:111-164   run_inference_impl (validation order and error texts kept)
//   engine.rs:199-263   run_inference_blob_impl
//   engine.rs:292-305   get_model_metadata_impl
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace infera_hip {

class LoadedModel;

struct OutShape {
  uint64_t len = 0, rows = 0, cols = 0;
};

std::pair<uint64_t, uint64_t> shape_rows_cols(const std::vector<uint64_t> &shape);

namespace engine {

// replaces an existing name; output_select: "" = first graph output, else an output's name or decimal index
void load_model(const std::string &name, const std::string &path, const std::string &output_select = "");
bool unload_model(const std::string &name);
std::shared_ptr<const LoadedModel> find(const std::string &name);  // throws ModelNotFound
std::vector<std::string> loaded_names();
std::string model_metadata_json(const std::string &name);

// Validation shared by every predict flavour (engine.rs:126-137 + the backend's input-fact check).
// Returns the output geometry for `rows` input rows.
OutShape validate_predict(const LoadedModel &m, uint64_t rows, uint64_t cols);
// Device-resident flavour: rank-2 models follow validate_predict; models with another input rank take
// `rows` samples of `cols` = prod(input_shape[1:]) elements each (the blob rule, engine.rs:233-238).
OutShape validate_device(const LoadedModel &m, uint64_t rows, uint64_t cols);
// Blob flavour (engine.rs:209-238): returns the batch (row) count.
uint64_t validate_blob(const LoadedModel &m, uint64_t blob_len);
OutShape out_shape_for_rows(const LoadedModel &m, uint64_t rows);

}  // namespace engine
}  // namespace infera_hip
