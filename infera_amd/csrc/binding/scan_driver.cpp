// scan_driver.cpp -- TEST / BENCH ONLY: table-scan drivers over the chunk ABI of sql_surface.h.  `threads` workers (DuckDB pipeline workers)
// pull 2048-row chunks of a columnar table in host memory from a shared counter and run `SELECT function(model, c1..cN)` on each through
// infera_sql_call -- which is provided by whatever this file is linked with: tests/duckdb_stub/driver.cpp, i.e. the REAL extension source
// (infera_extension_hip.cpp) behind the test-only DuckDB stand-in.  bench.py's `end_to_end` numbers are these scans (first
// chunk's gather -> last result element consumed).  Round 5 removed the second, mock implementation of the SQL functions that used to live
// beside these drivers (one SQL layer -- the file that ships is the file that is measured).
#include "sql_surface.h"

#include "infera.h"
#include "infera_hip.h"  // (infera_gather_columns_colmajor: the staged path's gather step alone)

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

}  // extern "C"  (a template cannot have C linkage)

namespace {
// `reps` complete scans of `rows` rows by `threads` workers pulling 2048-row chunks from a shared counter; `set_columns(chunk, rows in it, args + 1,
// scratch)` points the chunk's `ncols` argument vectors at the table (whatever its layout).  Shared by every table shape below.
struct ScanScratch {
  std::vector<float> vectors;  // a worker's own vector buffers (what DuckDB assembles a vector in when it cannot point into a segment)
};
template <class SetColumns>
int32_t run_table_scan(const char *function, const char *model, uint64_t rows, uint32_t ncols, int32_t threads, int32_t reps, double *secs,
                       double *checksum, char *err, uint64_t errlen, SetColumns &&set_columns) {
  if (threads < 1) threads = 1;
  const size_t CH = INFERA_SQL_VECTOR_SIZE;
  const uint64_t nchunks = (rows + CH - 1) / CH;
  const std::string fn = function ? function : "infera_predict";
  std::string first_error;
  std::mutex mu;
  // Work is dealt the way DuckDB's parallel table scan deals it: a worker claims a whole ROW GROUP (60 vectors) and walks it vector by vector,
  // so concurrent workers read different column segments (round 6: dealt chunk by chunk -- rounds 1-5 -- every worker walked the SAME 128
  // segment blocks at the same time, and the registered path's per-block reader counts bounced between their caches: 16 -> 9 us of "binding
  // layer" per chunk).  The last `threads` row groups are still dealt chunk by chunk so that every worker finishes within a chunk of the others:
  // a 10M-row table has only 82 row groups, and the idle tail of a 16-worker scan would otherwise be a property of the table's size, not of the
  // path (a 100M-row table has 814).  INFERA_BENCH_MORSEL=chunk: the old order.
  const uint64_t CPG = INFERA_SQL_ROW_GROUP / INFERA_SQL_VECTOR_SIZE, ngroups = (nchunks + CPG - 1) / CPG;
  const char *morsel_env = std::getenv("INFERA_BENCH_MORSEL");
  const bool by_group = !(morsel_env && morsel_env[0] == 'c');
  const uint64_t morsel_groups = by_group && ngroups > uint64_t(threads) ? ngroups - uint64_t(threads) : 0;
  for (int rep = 0; rep < reps; rep++) {
    std::atomic<uint64_t> next_group{0}, next{morsel_groups * CPG};
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
      ScanScratch scratch;
      const uint8_t *name_ptr = reinterpret_cast<const uint8_t *>(model);
      uint64_t name_len = std::strlen(model);
      args[0] = InferaSqlVector{INFERA_SQL_VARCHAR, 1, &name_ptr, &name_len, nullptr};
      double local = 0.0;
      uint64_t in_call = 0;
      const uint64_t t_thread0 = bench_now_ns();
      uint64_t gc = 0, gc_end = 0;  // the chunks of the row group this worker holds
      for (;;) {
        uint64_t c;
        if (gc < gc_end) {
          c = gc++;
        } else {
          const uint64_t g = next_group.load(std::memory_order_relaxed) < morsel_groups ? next_group.fetch_add(1, std::memory_order_relaxed) : morsel_groups;
          if (g < morsel_groups) {
            gc = g * CPG, gc_end = gc + CPG;
            continue;
          }
          c = next.fetch_add(1, std::memory_order_relaxed);
        }
        if (c >= nchunks) break;
        const size_t nr = size_t(std::min<uint64_t>(CH, rows - c * CH));
        set_columns(c, nr, args.data() + 1, scratch);
        InferaSqlResult res;
        const uint64_t t_c0 = bench_now_ns();
        const int32_t rc = infera_sql_call(fn.c_str(), args.data(), ncols + 1, nr, &res);
        in_call += bench_now_ns() - t_c0;
        if (rc != 0) {
          std::lock_guard<std::mutex> lk(mu);
          if (first_error.empty()) first_error = res.error ? res.error : "unknown error";
          infera_sql_free_result(&res);
          next.store(nchunks);
          next_group.store(morsel_groups);
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
}  // namespace

extern "C" {

int32_t infera_sql_bench_scan_table_typed(const char *function, const char *model, const void *table_v, int32_t elem_type, uint64_t rows,
                                          uint32_t ncols, int32_t threads, int32_t reps, double *secs, double *checksum, char *err,
                                          uint64_t errlen) {
  if (elem_type != INFERA_SQL_FLOAT && elem_type != INFERA_SQL_DOUBLE) {
    if (err && errlen) std::snprintf(err, size_t(errlen), "table element type must be FLOAT or DOUBLE");
    return -1;
  }
  const size_t esz = elem_type == INFERA_SQL_DOUBLE ? 8 : 4;
  const uint8_t *table = static_cast<const uint8_t *>(table_v);
  const uint64_t CH = INFERA_SQL_VECTOR_SIZE, RG = INFERA_SQL_ROW_GROUP;
  static_assert(INFERA_SQL_ROW_GROUP % INFERA_SQL_VECTOR_SIZE == 0, "chunks never straddle a row group");
  return run_table_scan(function, model, rows, ncols, threads, reps, secs, checksum, err, errlen,
                        [&](uint64_t c, size_t, InferaSqlVector *cols, ScanScratch &) {
                          const uint64_t row0 = c * CH, g = row0 / RG, g0 = g * RG, gr = std::min<uint64_t>(RG, rows - g0);
                          const uint8_t *base = table + (g0 * ncols + (row0 - g0)) * esz;
                          for (uint32_t j = 0; j < ncols; j++) cols[j] = InferaSqlVector{elem_type, 0, base + uint64_t(j) * gr * esz, nullptr, nullptr};
                        });
}

// ---- the HOST side of the staged scan alone (round 6) -----------------------------------------------------------------
// The same workers, the same dealing order, the same chunks of the same table -- but each chunk is only GATHERED (infera_gather_columns_colmajor:
// exactly what the staged path does into pinned staging, here into the worker's own 64-byte aligned buffer) and nothing is sent anywhere.  The
// aggregate rate at T threads is what the host's memory system delivers to staging buffers: eight staged GPUs need 8 x 56 GB/s of it (plus the
// DMA engines' reads of the same bytes), so `gb_per_s` / 56 is the SECOND bound of the 8-GPU model beside CPU time per chunk.
int32_t infera_sql_bench_gather_only(const float *table, uint64_t rows, uint32_t ncols, int32_t threads, int32_t reps, double *secs, double *cpu_seconds) {
  if (threads < 1) threads = 1;
  const uint64_t CH = INFERA_SQL_VECTOR_SIZE, RG = INFERA_SQL_ROW_GROUP, CPG = RG / CH;
  const uint64_t nchunks = (rows + CH - 1) / CH, ngroups = (nchunks + CPG - 1) / CPG;
  const uint64_t morsel_groups = ngroups > uint64_t(threads) ? ngroups - uint64_t(threads) : 0;
  std::atomic<int> failed{0};
  double cpu_total = 0.0;
  for (int rep = 0; rep < reps; rep++) {
    std::atomic<uint64_t> next_group{0}, next{morsel_groups * CPG};
    auto worker = [&] {
      float *buf = static_cast<float *>(std::aligned_alloc(4096, size_t(ncols) * CH * sizeof(float)));
      std::vector<infera::InferaColumn> cols(ncols);
      std::memset(buf, 0, size_t(ncols) * CH * sizeof(float));  // (pages touched before the clock matters: the staged path's buffers are long-lived)
      uint64_t gc = 0, gc_end = 0;
      for (;;) {
        uint64_t c;
        if (gc < gc_end) {
          c = gc++;
        } else {
          const uint64_t g = next_group.load(std::memory_order_relaxed) < morsel_groups ? next_group.fetch_add(1, std::memory_order_relaxed) : morsel_groups;
          if (g < morsel_groups) {
            gc = g * CPG, gc_end = gc + CPG;
            continue;
          }
          c = next.fetch_add(1, std::memory_order_relaxed);
        }
        if (c >= nchunks) break;
        const uint64_t row0 = c * CH, g = row0 / RG, g0 = g * RG, gr = std::min<uint64_t>(RG, rows - g0);
        const size_t nr = size_t(std::min<uint64_t>(CH, rows - row0));
        const float *base = table + g0 * ncols + (row0 - g0);
        for (uint32_t j = 0; j < ncols; j++) cols[j] = infera::InferaColumn{base + uint64_t(j) * gr, nullptr, infera::INFERA_COL_FLOAT, 0};
        if (infera::infera_gather_columns_colmajor(cols.data(), ncols, 0, nr, buf) != 0) failed = 1;
      }
      std::free(buf);
    };
    const double cpu0 = process_cpu_seconds();
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker);
    for (auto &x : th) x.join();
    if (secs) secs[rep] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    cpu_total += process_cpu_seconds() - cpu0;
  }
  if (cpu_seconds) *cpu_seconds = cpu_total;
  return failed ? -1 : 0;
}

// ---- the same table in DuckDB's SEGMENT shape (round 6) --------------------------------------------------------------
// What a scan's FLAT vectors point into when DuckDB holds an uncompressed FLOAT column: not one [columns][rows] matrix but column SEGMENTS,
// each in its own buffer-manager block of Storage::BLOCK_ALLOC_SIZE = 256 KiB taken from DBConfig::allocator, the first BLOCK_HEADER_SIZE = 8
// bytes of which are the block header -- so a segment holds (262144 - 8) / 4 = 65,534 values, a row group of 122,880 rows is two segments per
// column, the vectors of a segment lie 8 bytes past a 16-byte boundary of its block (those of a row group's SECOND segment start 2 values in:
// 16-byte aligned again), and the one vector per row group that straddles the two segments (rows 63,488..65,535) cannot be pointed at: DuckDB
// assembles it in the vector's own buffer (here: the worker's scratch, ordinary memory -> that chunk is staged).  (Constants as of DuckDB 1.x,
// quoted from memory -- external/duckdb is an empty submodule: INTEGRATION.md 2.2 marks them "unverified without a DuckDB tree".)
// 128 columns of one chunk = 128 unrelated block addresses: no 2-D copy can fetch them, every chunk is the pulling kernel's.
struct InferaSqlSegmentTable {
  uint64_t rows = 0, seg_values = 0, block_bytes = 0, header_bytes = 0;
  uint32_t ncols = 0;
  uint32_t segs_per_group = 0;
  std::vector<uint8_t *> blocks;  // [(group * segs_per_group + seg) * ncols + col]
  infera_sql_block_free_fn free_fn = nullptr;
  void *alloc_ctx = nullptr;
};

InferaSqlSegmentTable *infera_sql_segment_table_create(uint64_t rows, uint32_t ncols, uint64_t seed, int32_t threads, uint64_t block_bytes,
                                                       uint64_t header_bytes, infera_sql_block_alloc_fn alloc_fn, infera_sql_block_free_fn free_fn,
                                                       void *alloc_ctx, uint64_t shuffle_seed, int32_t alloc_threads) {
  if (!alloc_fn || !free_fn || block_bytes <= header_bytes + 4 || ncols == 0) return nullptr;
  if (threads < 1) threads = 1;
  auto *t = new InferaSqlSegmentTable;
  t->rows = rows, t->ncols = ncols, t->block_bytes = block_bytes, t->header_bytes = header_bytes;
  t->seg_values = (block_bytes - header_bytes) / 4;
  const uint64_t RG = INFERA_SQL_ROW_GROUP, ngroups = (rows + RG - 1) / RG;
  t->segs_per_group = uint32_t((RG + t->seg_values - 1) / t->seg_values);
  t->free_fn = free_fn, t->alloc_ctx = alloc_ctx;
  t->blocks.assign(size_t(ngroups) * t->segs_per_group * ncols, nullptr);
  // ALLOCATION ORDER decides where a chunk's 128 blocks lie relative to each other, and with an arena allocator whether they lie at one
  // stride: shuffle_seed = 0: a (row group, segment)'s columns back to back, group after group -- a table loaded by ONE thread; != 0: every
  // block of the table in a random order -- parallel loads, evictions and reloads interleaved with everything else the allocator serves:
  // 128 unrelated addresses per chunk whatever the allocator does (VERDICT r5 item 2's shape; the conservative row of every table).
  std::vector<size_t> order;
  for (uint64_t g = 0; g < ngroups; g++) {
    const uint64_t gr = std::min<uint64_t>(RG, rows - g * RG);
    for (uint32_t k = 0; k < t->segs_per_group; k++) {
      if (uint64_t(k) * t->seg_values >= gr) continue;
      for (uint32_t c = 0; c < ncols; c++) order.push_back((size_t(g) * t->segs_per_group + k) * ncols + c);
    }
  }
  if (shuffle_seed)
    for (size_t i = order.size(); i > 1; i--) std::swap(order[i - 1], order[size_t(splitmix64(shuffle_seed + i) % i)]);
  bool ok = true;
  if (alloc_threads > 1 && !shuffle_seed) {
    // a PARALLEL load: `alloc_threads` workers each claim a (row group, segment) at a time and allocate its `ncols` blocks one after the other
    // (DuckDB's parallel insert: every thread appends to row groups of its own) -- the allocator sees the threads' requests interleaved
    std::atomic<size_t> next_set{0};
    std::atomic<bool> failed{false};
    const size_t nsets = order.size() / ncols;
    std::vector<std::thread> workers;
    for (int w = 0; w < alloc_threads; w++)
      workers.emplace_back([&] {
        for (;;) {
          const size_t set = next_set.fetch_add(1);
          if (set >= nsets || failed.load()) break;
          for (uint32_t c = 0; c < ncols; c++) {
            void *b = alloc_fn(alloc_ctx, block_bytes);
            if (!b) {
              failed = true;
              break;
            }
            t->blocks[order[set * ncols + c]] = static_cast<uint8_t *>(b);
          }
        }
      });
    for (auto &x : workers) x.join();
    ok = !failed.load();
  } else {
    for (size_t slot : order) {
      void *b = alloc_fn(alloc_ctx, block_bytes);
      if (!b) {
        ok = false;
        break;
      }
      t->blocks[slot] = static_cast<uint8_t *>(b);
    }
  }
  if (!ok) {
    infera_sql_segment_table_destroy(t);
    return nullptr;
  }
  std::atomic<uint64_t> next{0};
  auto worker = [&] {
    for (;;) {
      const uint64_t task = next.fetch_add(1, std::memory_order_relaxed);  // one block per task
      if (task >= t->blocks.size()) break;
      uint8_t *b = t->blocks[size_t(task)];
      if (!b) continue;
      const uint64_t c = task % ncols, gk = task / ncols, g = gk / t->segs_per_group, k = gk % t->segs_per_group;
      const uint64_t gr = std::min<uint64_t>(RG, rows - g * RG), r0 = k * t->seg_values, n = std::min<uint64_t>(t->seg_values, gr - r0);
      std::memset(b, 0, size_t(header_bytes));
      float *dst = reinterpret_cast<float *>(b + header_bytes);
      for (uint64_t r = 0; r < n; r++) {
        const uint64_t u = splitmix64(seed ^ ((g * RG + r0 + r) * ncols + c));
        dst[r] = float(u >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
      }
    }
  };
  std::vector<std::thread> th;
  for (int i = 0; i < threads; i++) th.emplace_back(worker);
  for (auto &x : th) x.join();
  return t;
}

void infera_sql_segment_table_destroy(InferaSqlSegmentTable *t) {
  if (!t) return;
  for (uint8_t *b : t->blocks)
    if (b) t->free_fn(t->alloc_ctx, b, t->block_bytes);
  delete t;
}

uint64_t infera_sql_segment_table_blocks(const InferaSqlSegmentTable *t) {
  uint64_t n = 0;
  if (t)
    for (uint8_t *b : t->blocks) n += b != nullptr;
  return n;
}

int32_t infera_sql_bench_scan_segments(const char *function, const char *model, const InferaSqlSegmentTable *t, uint64_t rows, int32_t threads,
                                       int32_t reps, double *secs, double *checksum, uint64_t *assembled_chunks, char *err, uint64_t errlen) {
  if (!t || rows > t->rows) {
    if (err && errlen) std::snprintf(err, size_t(errlen), "no segment table / more rows than it holds");
    return -1;
  }
  const uint64_t CH = INFERA_SQL_VECTOR_SIZE, RG = INFERA_SQL_ROW_GROUP, SV = t->seg_values;
  const uint32_t ncols = t->ncols;
  std::atomic<uint64_t> assembled{0};
  const int32_t rc = run_table_scan(
      function, model, rows, ncols, threads, reps, secs, checksum, err, errlen, [&](uint64_t c, size_t nr, InferaSqlVector *cols, ScanScratch &scratch) {
        const uint64_t row0 = c * CH, g = row0 / RG, in_group = row0 - g * RG, k = in_group / SV, off = in_group - k * SV;
        uint8_t *const *blk = t->blocks.data() + (size_t(g) * t->segs_per_group + k) * ncols;
        if (off + nr <= SV) {  // the whole vector lies in one segment: a FLAT vector pointing into the block (zero-copy inside DuckDB too)
          for (uint32_t j = 0; j < ncols; j++)
            cols[j] = InferaSqlVector{INFERA_SQL_FLOAT, 0, blk[j] + t->header_bytes + off * 4, nullptr, nullptr};
          return;
        }
        // straddles two segments: assembled in the vector's own buffer, as DuckDB's scan does (two partial scans)
        assembled.fetch_add(1, std::memory_order_relaxed);
        scratch.vectors.resize(size_t(ncols) * CH);
        const uint64_t first = SV - off;
        uint8_t *const *blk2 = blk + ncols;
        for (uint32_t j = 0; j < ncols; j++) {
          float *v = scratch.vectors.data() + size_t(j) * CH;
          std::memcpy(v, blk[j] + t->header_bytes + off * 4, size_t(first) * 4);
          std::memcpy(v + first, blk2[j] + t->header_bytes, size_t(nr - first) * 4);
          cols[j] = InferaSqlVector{INFERA_SQL_FLOAT, 0, v, nullptr, nullptr};
        }
      });
  if (assembled_chunks) *assembled_chunks = assembled.load();
  return rc;
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
