// scan_driver.cpp -- TEST / BENCH ONLY: table-scan drivers over the chunk ABI of sql_surface.h.  `threads` workers (DuckDB pipeline workers)
// pull 2048-row chunks of a columnar table in host memory from a shared counter and run `SELECT function(model, c1..cN)` on each through
// infera_sql_call -- which is provided by whatever this file is linked with: tests/duckdb_stub/driver.cpp, i.e. the REAL extension source
// (infera_extension_hip.cpp) behind the test-only DuckDB stand-in.  bench.py's `end_to_end` numbers are these scans (first
// chunk's gather -> last result element consumed).  Round 5 removed the second, mock implementation of the SQL functions that used to live
// beside these drivers (one SQL layer -- the file that ships is the file that is measured).
#include "sql_surface.h"

#include <sched.h>
#include <sys/resource.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" {

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
