// duckdb.hpp -- TEST-ONLY stand-in for the handful of DuckDB C++ types that
// infera_amd/csrc/binding/infera_extension_hip.cpp touches.
//
// DuckDB's headers are not in the build image (the reference's external/duckdb is an empty submodule, no network), so
// the real extension source could otherwise not even be compile-checked.  This file is NOT DuckDB and is never shipped:
// it models just enough of the vector API -- flat / constant / dictionary vectors with validity masks, string and list
// vectors, DataChunk, ScalarFunction(-Set), ExtensionLoader -- with the semantics the DuckDB documentation gives them,
// so that tests/duckdb_stub/driver.cpp can load the extension, bind a call like DuckDB's binder would (overload by
// argument count and types, constant-NULL folding) and execute it on real buffers.  Names and signatures follow
// DuckDB's public headers so that the extension source compiles unchanged against either.
#pragma once

#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#ifndef DUCKDB_EXTENSION_API
#define DUCKDB_EXTENSION_API __attribute__((visibility("default")))
#endif

namespace duckdb {

using idx_t = uint64_t;
using validity_t = uint64_t;
using data_t = uint8_t;
using data_ptr_t = data_t *;
using sel_t = uint32_t;
using std::string;
using std::vector;
template <class T> using unique_ptr = std::unique_ptr<T>;
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T, class... A> unique_ptr<T> make_uniq(A &&...a) { return unique_ptr<T>(new T(std::forward<A>(a)...)); }
constexpr idx_t STANDARD_VECTOR_SIZE = 2048;

// ---- exceptions: what() carries the text DuckDB prints ("Invalid Input Error: ...") --------------------------------
class Exception : public std::runtime_error {
public:
  explicit Exception(const string &m) : std::runtime_error(m) {}
};
class InvalidInputException : public Exception {
public:
  explicit InvalidInputException(const string &m) : Exception("Invalid Input Error: " + m) {}
};

// ---- values ---------------------------------------------------------------------------------------------------------
struct string_t {
  const char *ptr = nullptr;
  uint32_t len = 0;
  string_t() = default;
  string_t(const char *p, uint32_t n) : ptr(p), len(n) {}
  const char *GetData() const { return ptr; }
  const char *GetDataUnsafe() const { return ptr; }
  idx_t GetSize() const { return len; }
  string GetString() const { return string(ptr, len); }
};
struct list_entry_t {
  uint64_t offset = 0, length = 0;
  list_entry_t() = default;
  list_entry_t(uint64_t o, uint64_t l) : offset(o), length(l) {}
};

enum class LogicalTypeId : uint8_t { INVALID, BOOLEAN, INTEGER, BIGINT, FLOAT, DOUBLE, DECIMAL, VARCHAR, BLOB, LIST };

class LogicalType {
public:
  static constexpr LogicalTypeId BOOLEAN = LogicalTypeId::BOOLEAN, INTEGER = LogicalTypeId::INTEGER, BIGINT = LogicalTypeId::BIGINT,
                                 FLOAT = LogicalTypeId::FLOAT, DOUBLE = LogicalTypeId::DOUBLE, VARCHAR = LogicalTypeId::VARCHAR,
                                 BLOB = LogicalTypeId::BLOB;
  LogicalType() = default;
  LogicalType(LogicalTypeId id) : id_(id) {}  // NOLINT: implicit, as in DuckDB
  static LogicalType LIST(const LogicalType &child) {
    LogicalType t(LogicalTypeId::LIST);
    t.child_ = std::make_shared<LogicalType>(child);
    return t;
  }
  static LogicalType DECIMAL(uint8_t width, uint8_t scale) {
    LogicalType t(LogicalTypeId::DECIMAL);
    t.width_ = width;
    t.scale_ = scale;
    return t;
  }
  LogicalTypeId id() const { return id_; }
  const LogicalType &child() const { return *child_; }
  uint8_t scale() const { return scale_; }
  bool operator==(const LogicalType &o) const {
    return id_ == o.id_ && (id_ != LogicalTypeId::LIST || *child_ == *o.child_) && (id_ != LogicalTypeId::DECIMAL || (width_ == o.width_ && scale_ == o.scale_));
  }
  bool operator!=(const LogicalType &o) const { return !(*this == o); }
  string ToString() const {
    switch (id_) {
      case LogicalTypeId::BOOLEAN: return "BOOLEAN";
      case LogicalTypeId::INTEGER: return "INTEGER";
      case LogicalTypeId::BIGINT: return "BIGINT";
      case LogicalTypeId::FLOAT: return "FLOAT";
      case LogicalTypeId::DOUBLE: return "DOUBLE";
      case LogicalTypeId::DECIMAL: return "DECIMAL(" + std::to_string(width_) + "," + std::to_string(scale_) + ")";
      case LogicalTypeId::VARCHAR: return "VARCHAR";
      case LogicalTypeId::BLOB: return "BLOB";
      case LogicalTypeId::LIST: return child_->ToString() + "[]";
      default: return "INVALID";
    }
  }
  size_t PhysicalSize() const {  // stub: DECIMAL is stored as int64
    switch (id_) {
      case LogicalTypeId::BOOLEAN: return 1;
      case LogicalTypeId::INTEGER: case LogicalTypeId::FLOAT: return 4;
      case LogicalTypeId::BIGINT: case LogicalTypeId::DOUBLE: case LogicalTypeId::DECIMAL: return 8;
      case LogicalTypeId::VARCHAR: case LogicalTypeId::BLOB: return sizeof(string_t);
      case LogicalTypeId::LIST: return sizeof(list_entry_t);
      default: return 1;
    }
  }

private:
  LogicalTypeId id_ = LogicalTypeId::INVALID;
  shared_ptr<LogicalType> child_;
  uint8_t width_ = 0, scale_ = 0;
};

// ---- validity / selection --------------------------------------------------------------------------------------------
class ValidityMask {
public:
  bool AllValid() const { return !mask_; }
  bool RowIsValid(idx_t i) const { return !mask_ || (((*mask_)[i >> 6] >> (i & 63)) & 1); }
  const validity_t *GetData() const { return mask_ ? mask_->data() : nullptr; }
  void SetInvalid(idx_t i) {
    Materialise(i + 1);
    (*mask_)[i >> 6] &= ~(validity_t(1) << (i & 63));
  }
  void SetValid(idx_t i) {
    if (mask_ && (i >> 6) < mask_->size()) (*mask_)[i >> 6] |= validity_t(1) << (i & 63);
  }
  void Reset() { mask_.reset(); }
  // stub-only: adopt an external word array (the driver wraps the test's buffers)
  void Adopt(const validity_t *words, idx_t rows) { mask_ = std::make_shared<vector<validity_t>>(words, words + (rows + 63) / 64); }

private:
  void Materialise(idx_t rows) {
    const size_t need = std::max<size_t>((rows + 63) / 64, STANDARD_VECTOR_SIZE / 64);
    if (!mask_) mask_ = std::make_shared<vector<validity_t>>(need, ~validity_t(0));
    else if (mask_->size() < need) mask_->resize(need, ~validity_t(0));
  }
  shared_ptr<vector<validity_t>> mask_;
};

class SelectionVector {
public:
  SelectionVector() = default;
  explicit SelectionVector(const sel_t *p) : sel_(p) {}
  bool IsSet() const { return sel_ != nullptr; }
  idx_t get_index(idx_t i) const { return sel_ ? sel_[i] : i; }
  const sel_t *data() const { return sel_; }

private:
  const sel_t *sel_ = nullptr;  // nullptr = incremental (0, 1, 2, ...)
};

enum class VectorType : uint8_t { FLAT_VECTOR, CONSTANT_VECTOR, DICTIONARY_VECTOR };

struct UnifiedVectorFormat {
  UnifiedVectorFormat() = default;
  UnifiedVectorFormat(const UnifiedVectorFormat &) = delete;  // as in DuckDB: neither copyable nor (portably) movable
  UnifiedVectorFormat &operator=(const UnifiedVectorFormat &) = delete;
  const SelectionVector *sel = nullptr;
  data_ptr_t data = nullptr;
  ValidityMask validity;
  SelectionVector owned_sel;  // constant vectors: every row maps to entry 0
  vector<sel_t> zeros;
  template <class T> static const T *GetData(const UnifiedVectorFormat &f) { return reinterpret_cast<const T *>(f.data); }
};

// ---- Vector ---------------------------------------------------------------------------------------------------------------
class Vector {
public:
  explicit Vector(LogicalType type, idx_t capacity = STANDARD_VECTOR_SIZE) : type_(std::move(type)) {
    owned_ = std::make_shared<vector<uint8_t>>(size_t(capacity) * type_.PhysicalSize(), uint8_t(0));
    data_ = owned_->data();
  }
  // stub-only: a FLAT vector over buffers the caller owns (how a scan hands DuckDB's own column buffers to a function)
  Vector(LogicalType type, data_ptr_t external) : type_(std::move(type)), data_(external) {}
  Vector(Vector &&) = default;
  Vector &operator=(Vector &&) = default;
  Vector(const Vector &) = delete;

  const LogicalType &GetType() const { return type_; }
  VectorType GetVectorType() const { return vtype_; }
  void SetVectorType(VectorType t) { vtype_ = t; }

  void ToUnifiedFormat(idx_t count, UnifiedVectorFormat &out) {
    out.data = data_;
    out.validity = validity_;
    if (vtype_ == VectorType::CONSTANT_VECTOR) {
      out.zeros.assign(size_t(std::max<idx_t>(count, 1)), 0);
      out.owned_sel = SelectionVector(out.zeros.data());
      out.sel = &out.owned_sel;
    } else if (vtype_ == VectorType::DICTIONARY_VECTOR) {
      out.owned_sel = SelectionVector(dict_sel_->data());
      out.sel = &out.owned_sel;
    } else {
      out.sel = &Incremental();
    }
  }
  // stub-only: turn this vector into a dictionary over its current buffer (row r reads entry sel[r]; validity is per entry)
  void MakeDictionary(vector<sel_t> sel) {
    dict_sel_ = std::make_shared<vector<sel_t>>(std::move(sel));
    vtype_ = VectorType::DICTIONARY_VECTOR;
  }
  void Verify(idx_t) {}

  static const SelectionVector &Incremental() {
    static const SelectionVector inc;
    return inc;
  }

  // internals reached by the helper structs below
  LogicalType type_;
  VectorType vtype_ = VectorType::FLAT_VECTOR;
  shared_ptr<vector<uint8_t>> owned_;
  data_ptr_t data_ = nullptr;
  ValidityMask validity_;
  shared_ptr<vector<sel_t>> dict_sel_;
  shared_ptr<std::deque<string>> heap_;  // strings added through StringVector::AddString
  unique_ptr<Vector> child_;             // LIST child
  idx_t list_size_ = 0, list_capacity_ = 0;
};

struct FlatVector {
  template <class T> static T *GetData(Vector &v) { return reinterpret_cast<T *>(v.data_); }
  template <class T> static const T *GetData(const Vector &v) { return reinterpret_cast<const T *>(v.data_); }
  static ValidityMask &Validity(Vector &v) { return v.validity_; }
  static void SetNull(Vector &v, idx_t i, bool is_null) {
    if (is_null) v.validity_.SetInvalid(i);
    else v.validity_.SetValid(i);
  }
  static const SelectionVector *IncrementalSelectionVector() { return &Vector::Incremental(); }
};
struct ConstantVector {
  template <class T> static T *GetData(Vector &v) { return reinterpret_cast<T *>(v.data_); }
  static void SetNull(Vector &v, bool is_null) {
    if (is_null) v.validity_.SetInvalid(0);
    else v.validity_.Reset();
  }
  static bool IsNull(const Vector &v) { return !v.validity_.RowIsValid(0); }
};
struct StringVector {
  static string_t AddString(Vector &v, const char *data, idx_t len) {
    if (!v.heap_) v.heap_ = std::make_shared<std::deque<string>>();
    v.heap_->emplace_back(data, size_t(len));
    return string_t(v.heap_->back().data(), uint32_t(len));
  }
  static string_t AddString(Vector &v, const string &s) { return AddString(v, s.data(), s.size()); }
  static string_t AddString(Vector &v, const char *s) { return AddString(v, s, std::strlen(s)); }
};
struct ListVector {
  static void Reserve(Vector &v, idx_t capacity) {
    if (!v.child_) {
      v.child_ = make_uniq<Vector>(v.type_.child(), std::max<idx_t>(capacity, STANDARD_VECTOR_SIZE));
      v.list_capacity_ = std::max<idx_t>(capacity, STANDARD_VECTOR_SIZE);
    } else if (capacity > v.list_capacity_) {
      idx_t cap = v.list_capacity_;
      while (cap < capacity) cap *= 2;
      v.child_->owned_->resize(size_t(cap) * v.child_->type_.PhysicalSize());
      v.child_->data_ = v.child_->owned_->data();
      v.list_capacity_ = cap;
    }
  }
  static Vector &GetEntry(Vector &v) {
    if (!v.child_) Reserve(v, STANDARD_VECTOR_SIZE);
    return *v.child_;
  }
  static void SetListSize(Vector &v, idx_t n) { v.list_size_ = n; }
  static idx_t GetListSize(const Vector &v) { return v.list_size_; }
  static list_entry_t *GetData(Vector &v) { return reinterpret_cast<list_entry_t *>(v.data_); }
};

// ---- DataChunk / function plumbing ----------------------------------------------------------------------------------
class DataChunk {
public:
  vector<Vector> data;
  idx_t size() const { return count_; }
  idx_t ColumnCount() const { return data.size(); }
  void SetCardinality(idx_t n) { count_ = n; }

private:
  idx_t count_ = 0;
};
struct ExpressionState {};

struct VectorOperations {
  // stub: the one cast the extension asks for (DECIMAL, stored as scaled int64 here, -> DOUBLE), any vector form
  static bool DefaultCast(Vector &source, Vector &result, idx_t count, bool = false) {
    if (source.GetType().id() != LogicalTypeId::DECIMAL || result.GetType().id() != LogicalTypeId::DOUBLE)
      throw Exception("Conversion Error: stub DefaultCast only knows DECIMAL -> DOUBLE");
    UnifiedVectorFormat f;
    source.ToUnifiedFormat(count, f);
    double div = 1.0;
    for (int i = 0; i < source.GetType().scale(); i++) div *= 10.0;
    const bool constant = source.GetVectorType() == VectorType::CONSTANT_VECTOR;
    result.SetVectorType(constant ? VectorType::CONSTANT_VECTOR : VectorType::FLAT_VECTOR);
    auto *dst = FlatVector::GetData<double>(result);
    const auto *src = UnifiedVectorFormat::GetData<int64_t>(f);
    for (idx_t r = 0; r < (constant ? 1 : count); r++) {
      const idx_t i = f.sel->get_index(r);
      if (!f.validity.RowIsValid(i)) FlatVector::SetNull(result, r, true);
      else dst[r] = double(src[i]) / div;
    }
    return true;
  }
};

using scalar_function_t = std::function<void(DataChunk &, ExpressionState &, Vector &)>;
enum class FunctionStability : uint8_t { CONSISTENT, VOLATILE };
enum class FunctionErrors : uint8_t { CANNOT_ERROR, CAN_THROW_RUNTIME_ERROR };

class ScalarFunction {
public:
  ScalarFunction(string name_p, vector<LogicalType> arguments_p, LogicalType return_type_p, scalar_function_t function_p)
      : name(std::move(name_p)), arguments(std::move(arguments_p)), return_type(std::move(return_type_p)), function(std::move(function_p)) {}
  void SetVolatile() { stability = FunctionStability::VOLATILE; }
  void SetFallible() { errors = FunctionErrors::CAN_THROW_RUNTIME_ERROR; }
  string name;
  vector<LogicalType> arguments;
  LogicalType return_type;
  scalar_function_t function;
  FunctionStability stability = FunctionStability::CONSISTENT;
  FunctionErrors errors = FunctionErrors::CANNOT_ERROR;
};
class ScalarFunctionSet {
public:
  explicit ScalarFunctionSet(string name_p) : name(std::move(name_p)) {}
  void AddFunction(ScalarFunction f) { functions.push_back(std::move(f)); }
  string name;
  vector<ScalarFunction> functions;
};

// duckdb/common/allocator.hpp: the allocator every buffer-manager block comes from (DBConfig::allocator)
struct PrivateAllocatorData {
  virtual ~PrivateAllocatorData() = default;
};
typedef data_ptr_t (*allocate_function_ptr_t)(PrivateAllocatorData *private_data, idx_t size);
typedef void (*free_function_ptr_t)(PrivateAllocatorData *private_data, data_ptr_t pointer, idx_t size);
typedef data_ptr_t (*reallocate_function_ptr_t)(PrivateAllocatorData *private_data, data_ptr_t pointer, idx_t old_size, idx_t size);
class Allocator {
public:
  Allocator() : Allocator(DefaultAllocate, DefaultFree, DefaultReallocate, nullptr) {}
  Allocator(allocate_function_ptr_t a, free_function_ptr_t f, reallocate_function_ptr_t r, unique_ptr<PrivateAllocatorData> pd)
      : allocate_(a), free_(f), reallocate_(r), private_data_(std::move(pd)) {}
  data_ptr_t AllocateData(idx_t size) { return allocate_(private_data_.get(), size); }
  void FreeData(data_ptr_t p, idx_t size) { if (p) free_(private_data_.get(), p, size); }
  data_ptr_t ReallocateData(data_ptr_t p, idx_t old_size, idx_t size) { return reallocate_(private_data_.get(), p, old_size, size); }
  static data_ptr_t DefaultAllocate(PrivateAllocatorData *, idx_t size) { return static_cast<data_ptr_t>(std::malloc(size)); }
  static void DefaultFree(PrivateAllocatorData *, data_ptr_t p, idx_t) { std::free(p); }
  static data_ptr_t DefaultReallocate(PrivateAllocatorData *, data_ptr_t p, idx_t, idx_t size) { return static_cast<data_ptr_t>(std::realloc(p, size)); }

private:
  allocate_function_ptr_t allocate_;
  free_function_ptr_t free_;
  reallocate_function_ptr_t reallocate_;
  unique_ptr<PrivateAllocatorData> private_data_;
};
// duckdb/main/config.hpp (the one member used here)
struct DBConfig {
  unique_ptr<Allocator> allocator;
};

class DatabaseInstance {
public:
  vector<ScalarFunction> catalog;  // every registered overload
};
class ExtensionLoader {
public:
  ExtensionLoader(DatabaseInstance &db_p, string name_p) : db(db_p), extension_name(std::move(name_p)) {}
  void RegisterFunction(ScalarFunction f) { db.catalog.push_back(std::move(f)); }
  void RegisterFunction(ScalarFunctionSet set) {
    for (auto &f : set.functions) db.catalog.push_back(std::move(f));
  }
  DatabaseInstance &db;
  string extension_name;
};
class Extension {
public:
  virtual ~Extension() = default;
  virtual void Load(ExtensionLoader &loader) = 0;
  virtual std::string Name() = 0;
  virtual std::string Version() const { return ""; }
};

}  // namespace duckdb
