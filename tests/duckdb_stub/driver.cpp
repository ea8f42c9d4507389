// driver.cpp -- TEST-ONLY: loads infera_amd/csrc/binding/infera_extension_hip.cpp (compiled against the stub duckdb.hpp
// next to this file) and drives it through the chunk ABI of infera_amd/csrc/binding/sql_surface.h: the Python tests that
// replay the reference's sqllogictests (tests/test_sql_surface.py) and bench.py's end-to-end scans
// (csrc/binding/scan_driver.cpp, linked into the same library) all run the REAL extension source.  What DuckDB itself would do around a scalar function is restated here in a few lines: overload
// lookup by name and argument count, constant-NULL folding (default NULL handling), exception -> error text.
//
// One deliberate difference from DuckDB's binder: typed argument vectors (INTEGER, BIGINT, DECIMAL, ...) are handed to
// the function as they are instead of being cast to the overload's FLOAT / DOUBLE parameters, so that every branch of
// the extension's gather is exercised.
// INFERA_STUB_DICTIONARY=1 wraps every flat numeric argument in a dictionary vector (permuted buffer + selection
// vector), the form a filter or a join hands to a function.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "duckdb.hpp"
#include "../../infera_amd/csrc/binding/sql_surface.h"

extern "C" void infera_duckdb_cpp_init(duckdb::ExtensionLoader &loader);
extern "C" bool infera_install_zero_copy_allocator(duckdb::DBConfig &config);

namespace {
using namespace duckdb;

DatabaseInstance &db() {
  static DatabaseInstance instance;
  static std::once_flag once;
  std::call_once(once, [] {
    ExtensionLoader loader(instance, "infera");
    infera_duckdb_cpp_init(loader);
  });
  return instance;
}

const char *type_name(const LogicalType &t) {
  switch (t.id()) {
    case LogicalTypeId::FLOAT: return "FLOAT";
    case LogicalTypeId::BOOLEAN: return "BOOLEAN";
    case LogicalTypeId::VARCHAR: return "VARCHAR";
    case LogicalTypeId::LIST: return "FLOAT[]";
    default: return "?";
  }
}

// test-only encoding of a DECIMAL argument vector: type = 8 | (scale << 8), data = int64 (the value times 10^scale)
constexpr int32_t kDecimalTag = 8;

LogicalType logical_of(int32_t t) {
  if ((t & 0xff) == kDecimalTag) return LogicalType::DECIMAL(18, uint8_t(t >> 8));
  switch (t) {
    case INFERA_SQL_VARCHAR: return LogicalType::VARCHAR;
    case INFERA_SQL_FLOAT: return LogicalType::FLOAT;
    case INFERA_SQL_DOUBLE: return LogicalType::DOUBLE;
    case INFERA_SQL_INTEGER: return LogicalType::INTEGER;
    case INFERA_SQL_BIGINT: return LogicalType::BIGINT;
    case INFERA_SQL_BLOB: return LogicalType::BLOB;
    case INFERA_SQL_BOOLEAN: return LogicalType::BOOLEAN;
    default: return LogicalType::LIST(LogicalType::FLOAT);
  }
}

bool arg_is_null(const InferaSqlVector &v, size_t row) {
  if (!v.validity) return false;
  const size_t r = v.is_constant ? 0 : row;
  return !((v.validity[r >> 6] >> (r & 63)) & 1);
}

struct Keep {  // buffers the chunk's vectors point into
  std::vector<std::vector<uint8_t>> bytes;
  std::vector<std::vector<string_t>> strings;
};

Vector make_vector(const InferaSqlVector &a, size_t rows, Keep &keep, bool dictionary) {
  const LogicalType type = logical_of(a.type);
  const size_t n = a.is_constant ? 1 : rows;
  if (a.type == INFERA_SQL_VARCHAR || a.type == INFERA_SQL_BLOB) {
    keep.strings.emplace_back(std::max<size_t>(n, 1));
    auto &s = keep.strings.back();
    auto ptrs = static_cast<const uint8_t *const *>(a.data);
    for (size_t i = 0; i < n; i++) s[i] = string_t(reinterpret_cast<const char *>(ptrs[i]), uint32_t(a.lens ? a.lens[i] : std::strlen(reinterpret_cast<const char *>(ptrs[i]))));
    Vector v(type, reinterpret_cast<data_ptr_t>(s.data()));
    if (a.validity) FlatVector::Validity(v).Adopt(a.validity, n);
    if (a.is_constant) v.SetVectorType(VectorType::CONSTANT_VECTOR);
    return v;
  }
  const size_t width = type.PhysicalSize();
  if (a.is_constant || !dictionary) {
    // zero-copy: the vector IS the caller's column buffer, like a scan handing a column segment to the function
    Vector v(type, const_cast<data_ptr_t>(static_cast<const uint8_t *>(a.data)));
    if (a.validity) FlatVector::Validity(v).Adopt(a.validity, n);
    if (a.is_constant) v.SetVectorType(VectorType::CONSTANT_VECTOR);
    return v;
  }
  // dictionary form: entry k of the buffer holds logical row (rows-1-k); sel[r] = rows-1-r
  keep.bytes.emplace_back(std::max<size_t>(rows, 1) * width);
  uint8_t *buf = keep.bytes.back().data();
  std::vector<sel_t> sel(rows);
  for (size_t r = 0; r < rows; r++) {
    std::memcpy(buf + (rows - 1 - r) * width, static_cast<const uint8_t *>(a.data) + r * width, width);
    sel[r] = sel_t(rows - 1 - r);
  }
  Vector v(type, buf);
  if (a.validity) {
    std::vector<validity_t> words((rows + 63) / 64, 0);
    for (size_t r = 0; r < rows; r++)
      if ((a.validity[r >> 6] >> (r & 63)) & 1) words[(rows - 1 - r) >> 6] |= validity_t(1) << ((rows - 1 - r) & 63);
    FlatVector::Validity(v).Adopt(words.data(), rows);
  }
  v.MakeDictionary(std::move(sel));
  return v;
}

void fill_result(Vector &result, const LogicalType &type, size_t rows, InferaSqlResult *out) {
  const bool constant = result.GetVectorType() == VectorType::CONSTANT_VECTOR;
  out->is_constant = constant ? 1 : 0;
  const size_t n = constant ? 1 : rows;
  auto &validity = FlatVector::Validity(result);
  if (!validity.AllValid()) {
    out->validity = static_cast<uint64_t *>(std::calloc(std::max<size_t>((n + 63) / 64, 1), sizeof(uint64_t)));
    for (size_t i = 0; i < n; i++)
      if (validity.RowIsValid(i)) out->validity[i >> 6] |= uint64_t(1) << (i & 63);
  }
  switch (type.id()) {
    case LogicalTypeId::FLOAT:
      out->type = INFERA_SQL_FLOAT;
      out->f32 = static_cast<float *>(std::malloc(std::max<size_t>(n, 1) * sizeof(float)));
      std::memcpy(out->f32, FlatVector::GetData<float>(result), n * sizeof(float));
      break;
    case LogicalTypeId::BOOLEAN:
      out->type = INFERA_SQL_BOOLEAN;
      out->boolean = static_cast<uint8_t *>(std::malloc(std::max<size_t>(n, 1)));
      for (size_t i = 0; i < n; i++) out->boolean[i] = FlatVector::GetData<bool>(result)[i] ? 1 : 0;
      break;
    case LogicalTypeId::VARCHAR: {
      out->type = INFERA_SQL_VARCHAR;
      out->strings = static_cast<char **>(std::calloc(std::max<size_t>(n, 1), sizeof(char *)));
      const auto *s = FlatVector::GetData<string_t>(result);
      for (size_t i = 0; i < n; i++) {
        if (!validity.RowIsValid(i)) continue;
        out->strings[i] = static_cast<char *>(std::malloc(s[i].GetSize() + 1));
        std::memcpy(out->strings[i], s[i].GetData(), s[i].GetSize());
        out->strings[i][s[i].GetSize()] = 0;
      }
      break;
    }
    default: {  // LIST<FLOAT>
      out->type = INFERA_SQL_LIST_FLOAT;
      if (constant) break;
      const auto *e = FlatVector::GetData<list_entry_t>(result);
      const float *child = FlatVector::GetData<float>(ListVector::GetEntry(result));
      size_t total = 0;
      for (size_t i = 0; i < rows; i++)
        if (validity.RowIsValid(i)) total += e[i].length;
      out->list_offsets = static_cast<uint64_t *>(std::malloc((rows + 1) * sizeof(uint64_t)));
      out->list_values = static_cast<float *>(std::malloc(std::max<size_t>(total, 1) * sizeof(float)));
      size_t at = 0;
      for (size_t i = 0; i < rows; i++) {
        out->list_offsets[i] = at;
        if (!validity.RowIsValid(i)) continue;
        if (e[i].offset + e[i].length > ListVector::GetListSize(result)) throw Exception("Internal Error: list entry outside the child vector");
        std::memcpy(out->list_values + at, child + e[i].offset, e[i].length * sizeof(float));
        at += e[i].length;
      }
      out->list_offsets[rows] = at;
    }
  }
}

// Overload lookup by name, argument count and (preferably exact) argument types.  DuckDB's binder does this ONCE per query, not per
// chunk: a scan calls with the same signature chunk after chunk, so the last binding of each thread is kept (2,058 catalog entries
// compared by name would otherwise cost every chunk of a benchmark scan ~10 us the extension never sees inside DuckDB).
const ScalarFunction *bind(const char *function, const InferaSqlVector *argv, size_t nargs) {
  struct Bound {
    std::string fn;
    std::vector<int32_t> types;
    const ScalarFunction *f = nullptr;
  };
  thread_local Bound last;
  if (last.f && last.types.size() == nargs && last.fn == function) {
    bool same = true;
    for (size_t i = 0; i < nargs && same; i++) same = last.types[i] == argv[i].type;
    if (same) return last.f;
  }
  const std::string fn = function;
  const ScalarFunction *chosen = nullptr;
  bool any = false;
  for (const auto &f : db().catalog) {
    if (f.name != fn) continue;
    any = true;
    if (f.arguments.size() != nargs) continue;
    bool exact = true;
    for (size_t i = 0; i < nargs; i++) exact = exact && f.arguments[i] == logical_of(argv[i].type);
    if (!chosen || exact) chosen = &f;
    if (exact) break;
  }
  if (!any) throw Exception("Catalog Error: Scalar Function with name " + fn + " does not exist!");
  if (!chosen)
    throw Exception("Binder Error: No function matches the given name and argument types '" + fn + "' with " + std::to_string(nargs) + " arguments");
  last.fn = fn;
  last.types.resize(nargs);
  for (size_t i = 0; i < nargs; i++) last.types[i] = argv[i].type;
  last.f = chosen;
  return chosen;
}

}  // namespace

extern "C" {

int32_t infera_sql_call(const char *function, const InferaSqlVector *argv, uintptr_t nargs, uintptr_t rows, InferaSqlResult *out) {
  std::memset(out, 0, sizeof *out);
  out->rows = rows;
  try {
    if (!function) throw InvalidInputException("function name is NULL");
    const ScalarFunction *chosen = bind(function, argv, nargs);
    if (rows > STANDARD_VECTOR_SIZE) throw InvalidInputException("chunk larger than STANDARD_VECTOR_SIZE");
    // default NULL handling: a constant NULL argument folds the call to a constant NULL (the body is never entered)
    if (rows > 0)
      for (size_t c = 0; c < nargs; c++)
        if (argv[c].is_constant && arg_is_null(argv[c], 0)) {
          out->type = chosen->return_type.id() == LogicalTypeId::FLOAT ? INFERA_SQL_FLOAT
                      : chosen->return_type.id() == LogicalTypeId::BOOLEAN ? INFERA_SQL_BOOLEAN
                      : chosen->return_type.id() == LogicalTypeId::VARCHAR ? INFERA_SQL_VARCHAR : INFERA_SQL_LIST_FLOAT;
          out->is_constant = 1;
          out->validity = static_cast<uint64_t *>(std::calloc(1, sizeof(uint64_t)));
          return 0;
        }
    const char *dict_env = std::getenv("INFERA_STUB_DICTIONARY");
    const bool dictionary = dict_env && *dict_env == '1';
    Keep keep;
    DataChunk chunk;
    for (size_t c = 0; c < nargs; c++) chunk.data.push_back(make_vector(argv[c], rows, keep, dictionary));
    chunk.SetCardinality(rows);
    Vector result(chosen->return_type, std::max<idx_t>(rows, 1));
    ExpressionState state;
    chosen->function(chunk, state, result);
    fill_result(result, chosen->return_type, rows, out);
    return 0;
  } catch (const std::exception &e) {
    infera_sql_free_result(out);
    out->status = -1;
    out->error = strdup(e.what());
    return -1;
  }
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
  struct Agg {
    size_t min_args = SIZE_MAX, max_args = 0, overloads = 0;
    std::string returns;
    bool is_volatile = false, fallible = false;
  };
  std::map<std::string, Agg> by_name;
  for (const auto &f : db().catalog) {
    Agg &a = by_name[f.name];
    a.min_args = std::min(a.min_args, f.arguments.size());
    a.max_args = std::max(a.max_args, f.arguments.size());
    a.overloads++;
    a.returns = type_name(f.return_type);
    a.is_volatile = f.stability == FunctionStability::VOLATILE;
    a.fallible = f.errors == FunctionErrors::CAN_THROW_RUNTIME_ERROR;
  }
  std::string o = "[";
  for (const auto &kv : by_name) {
    if (o.size() > 1) o += ",";
    o += "{\"name\":\"" + kv.first + "\",\"min_args\":" + std::to_string(kv.second.min_args) + ",\"max_args\":" + std::to_string(kv.second.max_args) +
         ",\"overloads\":" + std::to_string(kv.second.overloads) + ",\"returns\":\"" + kv.second.returns + "\",\"volatile\":" +
         (kv.second.is_volatile ? "true" : "false") + ",\"fallible\":" + (kv.second.fallible ? "true" : "false") + "}";
  }
  return strdup((o + "]").c_str());
}

// Cost of the extension's registration (256 x {FLOAT, DOUBLE} x 4 predict families = 2,048 overloads): builds a
// fresh catalog `reps` times and reports the seconds per load, the overloads registered and the LogicalType objects they hold
// (what a real DuckDB would copy into its catalog).  The stub's catalog is a vector, so this times the extension's side only.
double infera_stub_registration_seconds(int32_t reps, uint64_t *overloads, uint64_t *argument_types) {
  if (reps < 1) reps = 1;
  uint64_t n = 0, types = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) {
    DatabaseInstance fresh;
    ExtensionLoader loader(fresh, "infera");
    infera_duckdb_cpp_init(loader);
    n = fresh.catalog.size();
    types = 0;
    for (const auto &f : fresh.catalog) types += f.arguments.size();
  }
  if (overloads) *overloads = n;
  if (argument_types) *argument_types = types;
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
}

// The allocator hook end to end: `threads` workers each loop { allocate four 256 KiB blocks from DBConfig::allocator --
// the registering allocator when INFERA_ZERO_COPY_ALLOCATOR=1, malloc otherwise --, fill them with the 128 column runs of a 2048-row chunk,
// scan the chunk `scans_per_alloc` times through the extension's infera_predict (FLAT vectors pointing into the blocks, as a table scan
// hands them over), free the blocks } for `seconds`, all at once: allocation, scan and free of different threads overlap.  Every result is
// compared with the first result for that chunk content.  Returns rows/s; *mismatches / *allocs report what happened.
double infera_stub_allocator_scan(const char *model, int32_t threads, double seconds, int32_t scans_per_alloc, uint64_t *mismatches, uint64_t *allocs,
                                  int32_t *hook_installed, char *err, uint64_t errcap) {
  constexpr size_t K = 128, ROWS = 2048, BLOCK_COLS = 32, BLOCK = BLOCK_COLS * ROWS * sizeof(float);
  try {
    DBConfig config;
    const bool hooked = infera_install_zero_copy_allocator(config);
    if (!config.allocator) config.allocator = make_uniq<Allocator>();
    if (hook_installed) *hook_installed = hooked ? 1 : 0;
    const ScalarFunction *fn = nullptr;
    for (const auto &f : db().catalog)
      if (f.name == "infera_predict" && f.arguments.size() == K + 1 && f.arguments[1] == LogicalType::FLOAT) fn = &f;
    if (!fn) throw Exception("infera_predict with 128 FLOAT features is not registered");
    const std::string name = model;
    std::atomic<uint64_t> chunks{0}, bad{0}, nalloc{0};
    std::atomic<bool> stop{false};
    std::mutex err_mu;
    std::string first_error;
    auto worker = [&](int t) {
      std::vector<float> expected;
      for (uint64_t round = 0; !stop.load(std::memory_order_relaxed); round++) {
        data_ptr_t blk[K / BLOCK_COLS];
        for (auto &b : blk) b = config.allocator->AllocateData(BLOCK);
        nalloc.fetch_add(K / BLOCK_COLS, std::memory_order_relaxed);
        uint64_t s = 1234 + uint64_t(t);  // the same content every round: one expected result per thread
        for (size_t c = 0; c < K; c++) {
          float *run = reinterpret_cast<float *>(blk[c / BLOCK_COLS]) + (c % BLOCK_COLS) * ROWS;
          for (size_t r = 0; r < ROWS; r++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            run[r] = float(int32_t(s >> 40) - (1 << 23)) * (1.0f / float(1 << 23));
          }
        }
        try {
          for (int k = 0; k < scans_per_alloc && !stop.load(std::memory_order_relaxed); k++) {
            DataChunk chunk;
            string_t nm(name.data(), uint32_t(name.size()));
            Vector nv(LogicalType::VARCHAR, reinterpret_cast<data_ptr_t>(&nm));
            nv.SetVectorType(VectorType::CONSTANT_VECTOR);
            chunk.data.push_back(std::move(nv));
            for (size_t c = 0; c < K; c++)
              chunk.data.push_back(Vector(LogicalType::FLOAT, blk[c / BLOCK_COLS] + (c % BLOCK_COLS) * ROWS * sizeof(float)));
            chunk.SetCardinality(ROWS);
            Vector result(fn->return_type, ROWS);
            ExpressionState state;
            fn->function(chunk, state, result);
            const float *y = FlatVector::GetData<float>(result);
            if (expected.empty()) expected.assign(y, y + ROWS);
            else if (std::memcmp(expected.data(), y, ROWS * sizeof(float)) != 0) bad.fetch_add(1, std::memory_order_relaxed);
            chunks.fetch_add(1, std::memory_order_relaxed);
          }
        } catch (const std::exception &e) {
          std::lock_guard<std::mutex> lk(err_mu);
          if (first_error.empty()) first_error = e.what();
          stop.store(true);
        }
        for (auto &b : blk) config.allocator->FreeData(b, BLOCK);
      }
    };
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++) th.emplace_back(worker, t);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for (auto &x : th) x.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!first_error.empty()) throw Exception(first_error);
    if (mismatches) *mismatches = bad.load();
    if (allocs) *allocs = nalloc.load();
    return double(chunks.load()) * ROWS / sec;
  } catch (const std::exception &e) {
    if (err && errcap) std::snprintf(err, size_t(errcap), "%s", e.what());
    return -1.0;
  }
}

// A whole table in DuckDB's segment shape whose blocks come from DBConfig::allocator (round 6): the registering allocator of the extension when
// INFERA_ZERO_COPY_ALLOCATOR=1 (every 256 KiB block pinned where it lies as it is handed out -- what `DuckDB db(path, &config)` would do for
// the buffer manager's blocks), plain malloc otherwise.  The scans over it are csrc/binding/scan_driver.cpp's.  The allocator lives as long as
// the table (`*hook_installed` says which one it is).
namespace {
struct StubSegmentTable {
  DBConfig config;
  InferaSqlSegmentTable *table = nullptr;
};
void *stub_block_alloc(void *ctx, uint64_t bytes) { return static_cast<StubSegmentTable *>(ctx)->config.allocator->AllocateData(bytes); }
void stub_block_free(void *ctx, void *block, uint64_t bytes) {
  static_cast<StubSegmentTable *>(ctx)->config.allocator->FreeData(static_cast<data_ptr_t>(block), bytes);
}
}  // namespace

void *infera_stub_segment_table_create(uint64_t rows, uint32_t ncols, uint64_t seed, int32_t threads, int32_t *hook_installed, uint64_t shuffle_seed,
                                       int32_t alloc_threads) {
  auto *st = new StubSegmentTable;
  const bool hooked = infera_install_zero_copy_allocator(st->config);
  if (!st->config.allocator) st->config.allocator = make_uniq<Allocator>();
  if (hook_installed) *hook_installed = hooked ? 1 : 0;
  // Storage::BLOCK_ALLOC_SIZE = 262144, Storage::BLOCK_HEADER_SIZE = sizeof(uint64_t) (duckdb/storage/storage_info.hpp, quoted from memory)
  st->table = infera_sql_segment_table_create(rows, ncols, seed, threads, 262144, 8, stub_block_alloc, stub_block_free, st, shuffle_seed, alloc_threads);
  if (!st->table) {
    delete st;
    return nullptr;
  }
  return st;
}
const InferaSqlSegmentTable *infera_stub_segment_table_get(void *handle) { return handle ? static_cast<StubSegmentTable *>(handle)->table : nullptr; }
void infera_stub_segment_table_destroy(void *handle) {
  auto *st = static_cast<StubSegmentTable *>(handle);
  if (!st) return;
  infera_sql_segment_table_destroy(st->table);  // (frees every block through the allocator: the registering one unregisters it first)
  delete st;
}

// The allocator DBConfig::allocator holds after infera_install_zero_copy_allocator, driven call by call (tests/test_arena_allocator.py: runs
// without a GPU too -- a registration that fails only means the memory is plain memory).  One allocator per handle.
void *infera_stub_allocator_create(int32_t *hook_installed) {
  auto *config = new DBConfig;
  const bool hooked = infera_install_zero_copy_allocator(*config);
  if (!config->allocator) config->allocator = make_uniq<Allocator>();
  if (hook_installed) *hook_installed = hooked ? 1 : 0;
  return config;
}
void infera_stub_allocator_destroy(void *h) { delete static_cast<DBConfig *>(h); }
void *infera_stub_allocate(void *h, uint64_t size) { return static_cast<DBConfig *>(h)->allocator->AllocateData(size); }
void infera_stub_free(void *h, void *p, uint64_t size) { static_cast<DBConfig *>(h)->allocator->FreeData(static_cast<data_ptr_t>(p), size); }
void *infera_stub_reallocate(void *h, void *p, uint64_t old_size, uint64_t size) {
  return static_cast<DBConfig *>(h)->allocator->ReallocateData(static_cast<data_ptr_t>(p), old_size, size);
}

}  // extern "C"
