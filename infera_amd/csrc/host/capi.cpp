// capi.cpp -- the C ABI (include/infera.h + include/infera_hip.h).
//
// Each wrapper follows the shape of the reference's FFI functions in lib.rs: null checks ->
// UTF-8 check -> delegate -> map failure to (status | -1) + thread-local last error
// (lib.rs:38-64, 81-102, 127-149, 174-195, 215-233, 245-260, 275-285, 299-308, 326-366, 388-425).
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <optional>
#include <string>

#include "../../../include/infera_hip.h"
#include "../hip/backend.hpp"
#include "../hip/profile.hpp"
#include "common.hpp"
#include "engine.hpp"
#include "remote.hpp"
#include "gather.hpp"

using namespace infera_hip;

#include <atomic>

namespace {
std::atomic<uint64_t> g_zero_copy_calls{0};  // infera_predict_columns calls served from registered host memory


char *dup_cstr(const std::string &s) {
  char *p = static_cast<char *>(std::malloc(s.size() + 1));
  if (!p) return nullptr;
  std::memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

infera::InferaInferenceResult error_result() {  // ffi_utils.rs:28-36
  infera::InferaInferenceResult r;
  r.data = nullptr;
  r.len = r.rows = r.cols = 0;
  r.status = -1;
  return r;
}

std::string checked_str(const char *p) {
  if (!p) throw InferaError::null_pointer();
  if (!is_valid_utf8(p)) throw InferaError::utf8();
  return std::string(p);
}

// Runs fn, converting any exception into the last-error slot; returns true on success.
template <typename F>
bool guarded(F &&fn, std::string *err_text = nullptr) {
  try {
    fn();
    return true;
  } catch (const InferaError &e) {
    set_last_error(e.what());
    if (err_text) *err_text = e.what();
  } catch (const std::bad_alloc &) {
    set_last_error(InferaError::memory().what());
    if (err_text) *err_text = InferaError::memory().what();
  } catch (const std::exception &e) {
    std::string t = InferaError::onnx(e.what()).what();
    set_last_error(t);
    if (err_text) *err_text = t;
  }
  return false;
}

// An empty result carries a non-NULL, never-dereferenced, never-freed address -- what the reference hands out for
// an empty Box<[f32]> (a dangling aligned pointer with len 0, ffi_utils.rs:83-96) -- so infera_free_result must
// not pass a len-0 pointer to free(), whoever made it.
alignas(16) float g_empty_result[4];

float *alloc_out(uint64_t len) {
  if (len == 0) return g_empty_result;
  float *p = static_cast<float *>(std::malloc(len * sizeof(float)));
  if (!p) throw InferaError::memory();
  return p;
}
void free_out(float *p) {
  if (p != g_empty_result) std::free(p);
}

bool remove_tree(const std::string &dir) {
  DIR *d = ::opendir(dir.c_str());
  if (!d) return false;
  bool ok = true;
  while (dirent *e = ::readdir(d)) {
    const std::string n = e->d_name;
    if (n == "." || n == "..") continue;
    const std::string full = dir + "/" + n;
    struct stat st;
    if (::lstat(full.c_str(), &st) != 0) continue;
    ok = (S_ISDIR(st.st_mode) ? remove_tree(full) : ::unlink(full.c_str()) == 0) && ok;
  }
  ::closedir(d);
  return ::rmdir(dir.c_str()) == 0 && ok;
}

std::string error_json(const std::string &msg) { return "{\"error\":" + json_str(msg) + "}"; }

bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

bool is_regular_file(const std::string &p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

}  // namespace

namespace infera {
extern "C" {

int32_t infera_load_model(const char *name, const char *path) {
  return guarded([&] {
           if (!name || !path) throw InferaError::null_pointer();
           std::string n = checked_str(name), p = checked_str(path), select;
           // "<path>#<output>" serves another graph output than the first (name or index) -- additive, and never at the expense of
           // what the reference does with the same string:
           //  * a local path that exists as written, '#' included, is taken as written; the selector is split off only when the
           //    path before '#' exists (so a mistyped path is reported exactly as it was typed);
           //  * a URL is fetched and cached under the string AS WRITTEN (the reference's cache key is sha256 of that string,
           //    http.rs:187-190; the fragment never reaches the server -- the clients cut it from the request line), and its
           //    fragment selects an output only if the parsed model HAS an output of that name or index ("?" = optional:
           //    "http://h/m.onnx#v2" keeps loading output 0 like the reference).
           const size_t hash = p.rfind('#');
           const bool has_fragment = hash != std::string::npos && hash + 1 < p.size() && p.find('/', hash) == std::string::npos;
           if (p.rfind("http", 0) == 0) {
             if (has_fragment) select = "?" + p.substr(hash + 1);
             p = remote::handle_remote_model(p);  // lib.rs:47-51: fetch / revalidate into the cache
           } else if (has_fragment && ::access(p.c_str(), F_OK) != 0 && ::access(p.substr(0, hash).c_str(), F_OK) == 0) {
             select = p.substr(hash + 1);
             p.resize(hash);
           }
           engine::load_model(n, p, select);
         })
             ? 0
             : -1;
}

int32_t infera_unload_model(const char *name) {
  return guarded([&] {
           std::string n = checked_str(name);
           if (!engine::unload_model(n)) throw InferaError::model_not_found(n);
         })
             ? 0
             : -1;
}

struct InferaInferenceResult infera_predict(const char *model_name, const float *data, uintptr_t rows, uintptr_t cols) {
  InferaInferenceResult res = error_result();
  guarded([&] {
    if (!model_name || !data) throw InferaError::null_pointer();
    auto m = engine::find(checked_str(model_name));
    OutShape o = engine::validate_predict(*m, rows, cols);
    float *out = alloc_out(o.len);
    try {
      run_host(*m, data, out, int64_t(rows));
    } catch (...) {
      free_out(out);
      throw;
    }
    res.data = out;
    res.len = o.len;
    res.rows = o.rows;
    res.cols = o.cols;
    res.status = 0;
  });
  return res;
}

struct InferaInferenceResult infera_predict_from_blob(const char *model_name, const uint8_t *blob_data, uintptr_t blob_len) {
  InferaInferenceResult res = error_result();
  guarded([&] {
    if (!model_name || !blob_data) throw InferaError::null_pointer();
    auto m = engine::find(checked_str(model_name));
    const uint64_t rows = engine::validate_blob(*m, blob_len);
    OutShape o = engine::out_shape_for_rows(*m, rows);
    float *out = alloc_out(o.len);
    try {
      // native-endian bytes ARE the f32 row-major tensor (engine.rs:212-220); may be unaligned
      run_host(*m, reinterpret_cast<const float *>(blob_data), out, int64_t(rows));
    } catch (...) {
      free_out(out);
      throw;
    }
    res.data = out;
    res.len = o.len;
    res.rows = o.rows;
    res.cols = o.cols;
    res.status = 0;
  });
  return res;
}

char *infera_get_model_info(const char *model_name) {
  std::string json, err;
  if (guarded([&] { json = engine::model_metadata_json(checked_str(model_name)); }, &err)) return dup_cstr(json);
  return dup_cstr(error_json(err));  // lib.rs:225-232
}

char *infera_get_loaded_models(void) { return dup_cstr(json_str_array(engine::loaded_names())); }

char *infera_get_version(void) {
  // lib.rs:278-282; keys sorted as serde_json would emit them
  return dup_cstr("{\"model_cache_dir\":" + json_str(Config::get().cache_dir) +
                  ",\"onnx_backend\":\"hip-gfx950\",\"version\":\"0.4.0-mi355x\"}");
}

int32_t infera_clear_cache(void) {
  // http.rs clear_cache: remove the cache directory's contents; a missing directory is fine
  return guarded([&] {
           const std::string &dir = Config::get().cache_dir;
           DIR *d = ::opendir(dir.c_str());
           if (!d) return;
           while (dirent *e = ::readdir(d)) {
             std::string n = e->d_name;
             if (n == "." || n == "..") continue;
             std::string full = dir + "/" + n;
             struct stat st;
             if (::lstat(full.c_str(), &st) != 0) continue;
             // files and whole sub-directories (http.rs:132-138)
             const bool ok = S_ISDIR(st.st_mode) ? remove_tree(full) : ::unlink(full.c_str()) == 0;
             if (!ok) {
               ::closedir(d);
               throw InferaError::io("cannot remove " + full);
             }
           }
           ::closedir(d);
         })
             ? 0
             : -1;
}

char *infera_get_cache_info(void) {
  const Config &cfg = Config::get();
  uint64_t total = 0, count = 0;
  if (DIR *d = ::opendir(cfg.cache_dir.c_str())) {
    while (dirent *e = ::readdir(d)) {
      std::string full = cfg.cache_dir + "/" + e->d_name;
      struct stat st;
      if (ends_with(full, ".onnx") && ::stat(full.c_str(), &st) == 0 && S_ISREG(st.st_mode)) {
        total += uint64_t(st.st_size);
        count++;
      }
    }
    ::closedir(d);
  }
  // lib.rs:353-358, keys sorted
  return dup_cstr("{\"cache_dir\":" + json_str(cfg.cache_dir) + ",\"file_count\":" + std::to_string(count) +
                  ",\"size_limit_bytes\":" + std::to_string(cfg.cache_size_limit) + ",\"total_size_bytes\":" + std::to_string(total) + "}");
}

char *infera_set_autoload_dir(const char *path) {
  std::string err, json;
  bool ok = guarded(
      [&] {
        std::string dir = checked_str(path);
        DIR *d = ::opendir(dir.c_str());
        if (!d) throw InferaError::io(std::string(std::strerror(errno)) + " (" + dir + ")");
        std::vector<std::string> loaded;
        std::string errors;
        while (dirent *e = ::readdir(d)) {
          std::string fname = e->d_name;
          std::string full = dir + "/" + fname;
          if (!ends_with(fname, ".onnx") || !is_regular_file(full)) continue;
          std::string stem = fname.substr(0, fname.size() - 5);
          try {
            engine::load_model(stem, full);  // lib.rs:405-413 calls load_model_impl directly
            loaded.push_back(stem);
          } catch (const std::exception &ex) {
            if (!errors.empty()) errors += ",";
            errors += "{\"error\":" + json_str(ex.what()) + ",\"file\":" + json_str(full) + "}";
          }
        }
        ::closedir(d);
        json = "{\"errors\":[" + errors + "],\"loaded\":" + json_str_array(loaded) + "}";
      },
      &err);
  return dup_cstr(ok ? json : error_json(err));
}

const char *infera_last_error(void) { return last_error_cstr(); }

void infera_free(char *ptr) {
  if (ptr) std::free(ptr);
}

void infera_free_result(struct InferaInferenceResult res) {
  if (res.data && res.len) std::free(res.data);
}

// ================================ additive MI355X entry points ================================

char *infera_hip_sha256_hex(const char *data, uintptr_t len) {
  return dup_cstr(remote::sha256_hex(std::string(data ? data : "", data ? size_t(len) : 0)));
}

int32_t infera_hip_device_count(void) { return int32_t(devices().ids.size()); }

int32_t infera_hip_device_ordinal(int32_t i) {
  const auto &ds = devices();
  return (i >= 0 && size_t(i) < ds.ids.size()) ? ds.ids[size_t(i)] : -1;
}

char *infera_hip_get_devices(void) {
  const auto &ds = devices();
  std::string o = "{\"devices\":[";
  for (size_t i = 0; i < ds.ids.size(); i++) {
    if (i) o += ",";
    uint64_t calls = 0, rows = 0;
    slot_counters(int(i), &calls, &rows);
    std::string fault;
    const bool healthy = slot_health(int(i), &fault);
    o += "{\"arch\":" + json_str(ds.arch[i]) + ",\"cus\":" + std::to_string(ds.cus[i]) + (healthy ? "" : ",\"fault\":" + json_str(fault)) +
         ",\"healthy\":" + (healthy ? "true" : "false") + ",\"host_calls\":" + std::to_string(calls) +
         ",\"host_rows\":" + std::to_string(rows) + ",\"numa_node\":" + std::to_string(i < ds.numa.size() ? ds.numa[i] : -1) +
         ",\"ordinal\":" + std::to_string(ds.ids[i]) + ",\"pinned_staging_bytes\":" + std::to_string(slot_pinned_bytes(int(i))) + ",\"slot\":" + std::to_string(i) + "}";
  }
  o += "],\"host_phases\":" + host_phase_json() + ",\"registered_host_ranges\":" + std::to_string(registered_host_ranges()) + ",\"reason\":" + json_str(ds.why) + "}";
  return dup_cstr(o);
}

void infera_hip_shape_rows_cols(const uint64_t *shape, uintptr_t rank, uint64_t *rows, uint64_t *cols) {
  const auto rc = shape_rows_cols(std::vector<uint64_t>(shape, shape + (shape ? rank : 0)));
  if (rows) *rows = rc.first;
  if (cols) *cols = rc.second;
}

double infera_hip_h2d_probe(int32_t device, uint64_t bytes, int32_t iters, int32_t threads) {
  double r = -1.0;
  guarded([&] { r = h2d_probe_gbs(device, size_t(bytes), iters, threads); });
  return r;
}

int32_t infera_hip_choose_slot(const int32_t *slot_numa, uintptr_t nslots, int32_t thread_node, uint64_t ticket_on_node,
                               uint64_t ticket_global) {
  return choose_slot(std::vector<int>(slot_numa, slot_numa + (slot_numa ? nslots : 0)), thread_node, ticket_on_node, ticket_global);
}

int32_t infera_hip_register_host_memory(const void *base, uint64_t bytes) {
  return guarded([&] { register_host_memory(base, size_t(bytes)); }) ? 0 : -1;
}

int32_t infera_hip_unregister_host_memory(const void *base) {
  return guarded([&] {
           if (!base) throw InferaError::null_pointer();
           if (!unregister_host_memory(base)) throw InferaError::onnx("host memory range was not registered");
         })
             ? 0
             : -1;
}

uint64_t infera_hip_zero_copy_calls(void) { return g_zero_copy_calls.load(std::memory_order_relaxed); }

int32_t infera_hip_choose_slot_balanced(const int32_t *slot_numa, const int32_t *slot_threads, uintptr_t nslots, int32_t thread_node) {
  if (!slot_numa || !slot_threads) return 0;
  return choose_slot_balanced(std::vector<int>(slot_numa, slot_numa + nslots), std::vector<int>(slot_threads, slot_threads + nslots), thread_node);
}

char *infera_hip_get_plan(const char *model_name) {
  std::string json, err;
  if (guarded([&] { json = engine::find(checked_str(model_name))->describe_json(); }, &err)) return dup_cstr(json);
  return dup_cstr(error_json(err));
}

int32_t infera_hip_predict_device(const char *model_name, int32_t device, const float *d_in, uint64_t rows, uint64_t cols,
                                  float *d_out, uint64_t out_capacity, uint64_t *out_rows, uint64_t *out_cols) {
  return guarded([&] {
           if (!model_name || !d_in || !d_out) throw InferaError::null_pointer();
           auto m = engine::find(checked_str(model_name));
           OutShape o = engine::validate_device(*m, rows, cols);
           if (o.len > out_capacity)
             throw InferaError::onnx("output buffer too small: need " + std::to_string(o.len) + " elements, have " + std::to_string(out_capacity));
           run_device(*m, device, d_in, d_out, int64_t(rows));
           if (out_rows) *out_rows = o.rows;
           if (out_cols) *out_cols = o.cols;
         })
             ? 0
             : -1;
}

int32_t infera_hip_sync(int32_t device) { return guarded([&] { sync_device(device); }) ? 0 : -1; }

int32_t infera_hip_time_predict_device(const char *model_name, int32_t device, const float *d_in, uint64_t rows,
                                       uint64_t cols, float *d_out, uint64_t out_capacity, int32_t iters, float *elapsed_ms) {
  return guarded([&] {
           if (!model_name || !d_in || !d_out || !elapsed_ms) throw InferaError::null_pointer();
           auto m = engine::find(checked_str(model_name));
           OutShape o = engine::validate_device(*m, rows, cols);
           if (o.len > out_capacity) throw InferaError::onnx("output buffer too small");
           hipStream_t s = thread_stream(device);
           hipEvent_t e0, e1;
           if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) throw InferaError::onnx("HIP: hipEventCreate failed");
           (void)hipEventRecord(e0, s);
           for (int i = 0; i < iters; i++) run_device(*m, device, d_in, d_out, int64_t(rows));
           (void)hipEventRecord(e1, s);
           hipError_t e = hipEventSynchronize(e1);
           float ms = 0.f;
           if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
           (void)hipEventDestroy(e0);
           (void)hipEventDestroy(e1);
           if (e != hipSuccess) throw InferaError::onnx(std::string("HIP: event timing failed: ") + hipGetErrorString(e));
           *elapsed_ms = ms;
         })
             ? 0
             : -1;
}

void *infera_hip_malloc(int32_t device, uint64_t bytes) {
  void *p = nullptr;
  guarded([&] {
    if (hipSetDevice(device) != hipSuccess) throw InferaError::onnx("HIP: hipSetDevice failed");
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      p = nullptr;
      throw InferaError::onnx(std::string("HIP: hipMalloc: ") + hipGetErrorString(e));
    }
  });
  return p;
}

int32_t infera_hip_free(int32_t device, void *ptr) {
  return guarded([&] {
           if (!ptr) return;
           if (hipSetDevice(device) != hipSuccess || hipFree(ptr) != hipSuccess) throw InferaError::onnx("HIP: hipFree failed");
         })
             ? 0
             : -1;
}

int32_t infera_hip_memcpy_h2d(int32_t device, void *dst, const void *src, uint64_t bytes) {
  return guarded([&] {
           if (!dst || !src) throw InferaError::null_pointer();
           hipStream_t st = thread_stream(device);  // explicit stream: never the legacy stream (see hip/model.cpp upload)
           hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
           if (e == hipSuccess) e = hipStreamSynchronize(st);
           if (e != hipSuccess) throw InferaError::onnx(std::string("HIP: hipMemcpy H2D: ") + hipGetErrorString(e));
         })
             ? 0
             : -1;
}

int32_t infera_hip_memcpy_d2h(int32_t device, void *dst, const void *src, uint64_t bytes) {
  return guarded([&] {
           if (!dst || !src) throw InferaError::null_pointer();
           hipStream_t st = thread_stream(device);
           hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
           if (e == hipSuccess) e = hipStreamSynchronize(st);
           if (e != hipSuccess) throw InferaError::onnx(std::string("HIP: hipMemcpy D2H: ") + hipGetErrorString(e));
         })
             ? 0
             : -1;
}

int32_t infera_hip_synth_fill(int32_t device, float *d_dst, uint64_t seed, uint64_t row0, uint64_t rows, uint64_t cols) {
  return guarded([&] {
           if (!d_dst) throw InferaError::null_pointer();
           hipStream_t s = thread_stream(device);
           kern::synth_fill(s, d_dst, seed, row0, rows, cols);
           hipError_t e = hipGetLastError();
           if (e == hipSuccess) e = hipStreamSynchronize(s);
           if (e != hipSuccess) throw InferaError::onnx(std::string("HIP: synth_fill: ") + hipGetErrorString(e));
         })
             ? 0
             : -1;
}

int32_t infera_predict_into(const char *model_name, const float *data, uint64_t rows, uint64_t cols, float *out,
                            uint64_t out_capacity, uint64_t *out_rows, uint64_t *out_cols) {
  return guarded([&] {
           if (!model_name || !data || !out) throw InferaError::null_pointer();
           auto m = engine::find(checked_str(model_name));
           OutShape o = engine::validate_predict(*m, rows, cols);
           if (o.len > out_capacity)
             throw InferaError::onnx("output buffer too small: need " + std::to_string(o.len) + " elements, have " + std::to_string(out_capacity));
           run_host(*m, data, out, int64_t(rows));
           if (out_rows) *out_rows = o.rows;
           if (out_cols) *out_cols = o.cols;
         })
             ? 0
             : -1;
}

struct InferaInferenceResult infera_predict_columns(const char *model_name, const InferaColumn *columns, uintptr_t ncols,
                                                    uintptr_t rows) {
  InferaInferenceResult res = error_result();
  guarded([&] {
    if (!model_name || !columns) throw InferaError::null_pointer();
    std::optional<prof::Section> sec_find(std::in_place, 0, "capi: registry read + validation");
    auto m = engine::find(checked_str(model_name));
    // A NULL cell or an unsupported type must fail before any model-level error, as ExtractFeatures
    // runs before the FFI call in the reference (infera_extension.cpp:267-270).
    for (uintptr_t c = 0; c < ncols; c++) {
      const InferaColumn &col = columns[c];
      if (!col.data) throw InferaError::null_pointer();
      if (col.type < INFERA_COL_FLOAT || col.type > INFERA_COL_BIGINT)
        throw InferaError(ErrKind::Onnx, "Unsupported feature type: " + std::to_string(col.type));
      if (col.validity) {
        const uintptr_t n = col.is_constant ? (rows ? 1 : 0) : rows;
        for (uintptr_t r = 0; r < n; r++)
          if (!((col.validity[r >> 6] >> (r & 63)) & 1)) throw InferaError(ErrKind::Onnx, "Feature values cannot be NULL");
      }
    }
    OutShape o = engine::validate_predict(*m, rows, ncols);
    sec_find.reset();
    std::optional<prof::Section> sec_alloc(std::in_place, 1, "capi: result allocation");
    float *out = alloc_out(o.len);
    sec_alloc.reset();
    try {
      // Column-major staging: every column is converted / copied into pinned staging as the contiguous run it already is
      // (FLOAT: memcpy; DOUBLE / INTEGER: vectorised conversion of the run; constant vectors filled) and the GPU reads the
      // chunk column-major -- the host does 128 streaming passes per chunk instead of a 2048 x 128 transposing gather
      // (SURVEY.md 7 "hard parts": the host gather is what limits an 8-GPU host).  Plans whose first kernel cannot read a
      // column-major chunk get a GPU transpose in front; only hipGraph mode on such a plan (the transpose path allocates per
      // pass, which a capture cannot contain) falls back to the AVX2 transposing gather into row-major staging.
      // Zero-copy (round 3): when EVERY column run lies in host memory the application registered (infera_hip_register_host_memory),
      // the GPU reads the runs in place over PCIe and converts them on the way -- no CPU gather, no pinned staging, no H2D enqueue:
      // what a chunk costs the host drops from ~78 us of CPU to the launch and the wait.
      bool served = false;
      if (ncols > 0 && ncols <= uintptr_t(kern::kMaxZeroCopyCols) && registered_host_ranges() > 0 && Config::get().host_zero_copy) {
        std::optional<prof::Section> sec_lookup(std::in_place, 2, "capi: run table + lookup + pins");
        kern::ColumnTable tab;
        size_t esz[kern::kMaxZeroCopyCols], run_bytes[kern::kMaxZeroCopyCols];
        const void *host_ptr[kern::kMaxZeroCopyCols];
        for (uintptr_t c = 0; c < ncols; c++) {
          const InferaColumn &col = columns[c];
          esz[c] = col.type == INFERA_COL_FLOAT || col.type == INFERA_COL_INTEGER ? 4 : 8;
          host_ptr[c] = col.data;
          run_bytes[c] = (col.is_constant ? 1 : size_t(rows)) * esz[c];
          tab.type[c] = static_cast<unsigned char>(int(col.type) | (col.is_constant ? 8 : 0));
        }
        ZeroCopyPins pins;  // the page blocks under the runs stay mapped until this call has finished with them (unregistering waits for it)
        const bool all = lookup_host_memory_many(ncols, host_ptr, run_bytes, tab.ptr, pins);
        sec_lookup.reset();
        // FLOAT runs at ONE stride (a [columns][rows] matrix: a numpy / Arrow table, a row group whose columns were allocated together): the
        // chunk is a pitched rectangle, and ONE 2-D copy on the copy engines moves it -- they read pinned host memory at the link's DMA rate
        // (~55 GB/s), where a kernel pulling the same runs reaches 42 (round 3).  Anything else (typed columns, constants, blocks of a
        // database allocator scattered over the heap) keeps the pulling kernel.
        int64_t pitch = 0;
        // (the whole rectangle must lie inside ONE pinned block: the runtime resolves a 2-D copy's source to a single registration)
        bool rect = all && ncols >= 2 && pins.count == 1 && zero_copy_rect_enabled();
        for (uintptr_t c = 0; rect && c < ncols; c++) {
          rect = columns[c].type == INFERA_COL_FLOAT && !columns[c].is_constant;
          if (rect && c > 0) {
            const int64_t d = static_cast<const char *>(host_ptr[c]) - static_cast<const char *>(host_ptr[c - 1]);
            rect = d >= int64_t(rows) * 4 && (c == 1 || d == pitch);
            pitch = d;
          }
        }
        struct RectTicket {  // (held until the call returns: the copy is "in flight" until its chunk has been waited for)
          int t = -1;
          ~RectTicket() { rect_copy_release(t); }
        } ticket;
        if (rect) rect = (ticket.t = rect_copy_acquire()) >= 0;
        if (all)
          served = run_host_device_fill(*m, [&](hipStream_t stream, float *dst, int64_t r0, int64_t nr) {
            if (rect) {
              copy_rect_to_device(stream, dst, static_cast<const char *>(host_ptr[0]) + size_t(r0) * 4, size_t(pitch), size_t(nr) * 4, size_t(ncols));
              return;
            }
            // (Measured and dropped, round 3: ONE hipMemcpyBatchAsync of the 128 runs on the copy engines instead of this pulling kernel --
            // 292 us of CPU inside the call and 8.7 GB/s: the runtime issues 128 separate copies.)
            kern::ColumnTable t = tab;
            if (r0)
              for (uintptr_t c = 0; c < ncols; c++)
                if (!(t.type[c] & 8)) t.ptr[c] = static_cast<const char *>(t.ptr[c]) + size_t(r0) * esz[c];
            kern::gather_columns_device(stream, t, int(ncols), nr, dst);
          }, out, int64_t(rows));
        if (served) g_zero_copy_calls.fetch_add(1, std::memory_order_relaxed);
      }
      bool col_major = ncols > 0 && (colmajor_direct_ok(*m, int64_t(rows)) || !Config::get().use_hipgraph);
      if (served) {
        // (done: the GPU gathered the registered columns itself)
      } else if (col_major) {
        run_host_fill(*m, [&](float *dst, int64_t r0, int64_t nr) { gather_column_major(columns, 0, ncols, size_t(r0), size_t(nr), dst); },
                      out, int64_t(rows), /*col_major=*/true);
      } else {
        run_host_fill(*m, [&](float *dst, int64_t r0, int64_t nr) { gather_columns(columns, ncols, size_t(r0), size_t(nr), dst); }, out,
                      int64_t(rows));
      }
    } catch (...) {
      free_out(out);
      throw;
    }
    res.data = out;
    res.len = o.len;
    res.rows = o.rows;
    res.cols = o.cols;
    res.status = 0;
  });
  return res;
}

int32_t infera_gather_columns(const InferaColumn *columns, uintptr_t ncols, uintptr_t row0, uintptr_t nrows, float *dst) {
  return guarded([&] {
           if (!columns || !dst) throw InferaError::null_pointer();
           for (uintptr_t c = 0; c < ncols; c++) {
             const InferaColumn &col = columns[c];
             if (!col.data) throw InferaError::null_pointer();
             if (col.type < INFERA_COL_FLOAT || col.type > INFERA_COL_BIGINT)
               throw InferaError(ErrKind::Onnx, "Unsupported feature type: " + std::to_string(col.type));
             if (col.validity)
               for (uintptr_t r = col.is_constant ? 0 : row0; r < (col.is_constant ? (nrows ? 1 : 0) : row0 + nrows); r++)
                 if (!((col.validity[r >> 6] >> (r & 63)) & 1)) throw InferaError(ErrKind::Onnx, "Feature values cannot be NULL");
           }
           gather_columns(columns, ncols, row0, nrows, dst);
         })
             ? 0
             : -1;
}

int32_t infera_gather_columns_colmajor(const InferaColumn *columns, uintptr_t ncols, uintptr_t row0, uintptr_t nrows, float *dst) {
  return guarded([&] {
           if (!columns || !dst) throw InferaError::null_pointer();
           for (uintptr_t c = 0; c < ncols; c++) {
             const InferaColumn &col = columns[c];
             if (!col.data) throw InferaError::null_pointer();
             if (col.type < INFERA_COL_FLOAT || col.type > INFERA_COL_BIGINT)
               throw InferaError(ErrKind::Onnx, "Unsupported feature type: " + std::to_string(col.type));
             if (col.validity)
               for (uintptr_t r = col.is_constant ? 0 : row0; r < (col.is_constant ? (nrows ? 1 : 0) : row0 + nrows); r++)
                 if (!((col.validity[r >> 6] >> (r & 63)) & 1)) throw InferaError(ErrKind::Onnx, "Feature values cannot be NULL");
           }
           gather_column_major(columns, 0, ncols, row0, nrows, dst);
         })
             ? 0
             : -1;
}

struct InferaInferenceResult infera_predict_from_blob_batch(const char *model_name, const uint8_t *const *blobs,
                                                            const uintptr_t *lens, uintptr_t n) {
  InferaInferenceResult res = error_result();
  guarded([&] {
    if (!model_name || !blobs || !lens) throw InferaError::null_pointer();
    auto m = engine::find(checked_str(model_name));
    const auto &in = m->plan.input_shape;
    uint64_t per_sample = 1;
    for (size_t i = 1; i < in.size(); i++) per_sample *= uint64_t(in[i]);
    if (in.empty() || in[0] != -1)
      throw InferaError::onnx("batched BLOB inference needs a model with a symbolic leading (batch) dimension");
    for (uintptr_t i = 0; i < n; i++) {
      if (!blobs[i]) throw InferaError::null_pointer();
      if (lens[i] % 4 != 0) throw InferaError::invalid_blob_size();
      if (lens[i] / 4 != per_sample) throw InferaError::blob_shape_mismatch(size_t(per_sample), size_t(lens[i] / 4));
    }
    OutShape o = engine::out_shape_for_rows(*m, n);
    float *out = alloc_out(o.len);
    try {
      run_host_fill(*m,
                    [&](float *dst, int64_t r0, int64_t nr) {
                      for (int64_t i = 0; i < nr; i++) std::memcpy(dst + size_t(i) * per_sample, blobs[size_t(r0 + i)], size_t(per_sample) * 4);
                    },
                    out, int64_t(n));
    } catch (...) {
      free_out(out);
      throw;
    }
    res.data = out;
    res.len = o.len;
    res.rows = o.rows;
    res.cols = o.cols;
    res.status = 0;
  });
  return res;
}

}  // extern "C"
}  // namespace infera
