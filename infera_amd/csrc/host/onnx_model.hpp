// onnx_model.hpp -- in-memory form of the ONNX ModelProto subset the backend understands, and
// the protobuf wire decoder that fills it.
//
// Replaces what `tract_onnx::onnx().model_for_path(path)` does for the reference
// (engine.rs:49-51); the decoder is hand-written because neither libprotobuf headers for
// onnx.proto nor the `onnx` package exist in the build image.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace infera_hip::onnx {

enum DataType : int { kFloat = 1, kInt32 = 6, kInt64 = 7, kDouble = 11 };

struct TensorData {
  std::string name;
  int dtype = 0;  // kFloat or kInt64 after decoding (int32 widened, double narrowed)
  std::vector<int64_t> dims;
  std::vector<float> f32;
  std::vector<int64_t> i64;
  // Element count = payload length.  The decoder has verified it equals Π dims (overflow-checked), so a
  // declared shape can never claim more elements than the file holds.
  size_t count() const { return dtype == kInt64 ? i64.size() : f32.size(); }
};

struct Attribute {
  std::string name;
  int type = 0;  // AttributeProto.AttributeType: 1 FLOAT 2 INT 3 STRING 4 TENSOR 6 FLOATS 7 INTS
  float f = 0.f;
  int64_t i = 0;
  std::string s;
  std::vector<int64_t> ints;
  std::vector<float> floats;
  std::shared_ptr<TensorData> t;
};

struct NodeDef {
  std::string op, name, domain;
  std::vector<std::string> inputs, outputs;
  std::map<std::string, Attribute> attrs;

  int64_t attr_i(const std::string &k, int64_t dflt) const {
    auto it = attrs.find(k);
    return it == attrs.end() ? dflt : it->second.i;
  }
  float attr_f(const std::string &k, float dflt) const {
    auto it = attrs.find(k);
    return it == attrs.end() ? dflt : it->second.f;
  }
  std::string attr_s(const std::string &k, const std::string &dflt) const {
    auto it = attrs.find(k);
    return it == attrs.end() ? dflt : it->second.s;
  }
  const std::vector<int64_t> *attr_ints(const std::string &k) const {
    auto it = attrs.find(k);
    return it == attrs.end() ? nullptr : &it->second.ints;
  }
  const Attribute *attr(const std::string &k) const {
    auto it = attrs.find(k);
    return it == attrs.end() ? nullptr : &it->second;
  }
};

struct ValueDef {
  std::string name;
  int elem_type = 0;
  bool has_shape = false;
  std::vector<int64_t> dims;  // -1 = symbolic (dim_param) or unknown
};

struct Model {
  int64_t ir_version = 0;
  int64_t opset = 1;  // default-domain ("" / "ai.onnx") opset
  std::string producer, graph_name;
  std::vector<NodeDef> nodes;
  std::map<std::string, std::shared_ptr<TensorData>> initializers;
  std::vector<ValueDef> inputs;  // graph.input minus initializers
  std::vector<ValueDef> outputs;
};

// Throws InferaError::onnx(...) on I/O or wire-format problems.
Model parse_file(const std::string &path);
Model parse_bytes(const uint8_t *data, size_t len);

}  // namespace infera_hip::onnx
