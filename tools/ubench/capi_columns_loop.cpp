// capi_columns_loop.cpp -- (round 6) the ENGINE alone under the registered scan: T threads call infera_predict_columns on 2048-row chunks of a
// registered columnar host table directly -- no DuckDB stand-in, no extension source, no result consumer -- to separate what the C ABI and the
// host path cost per chunk from what the binding layer above them adds (tools/r06_duckdb_blocks.py measures the whole stack).
// build: g++ -std=c++17 -O2 -Iinclude -o tools/ubench/capi_columns_loop tools/ubench/capi_columns_loop.cpp -Linfera_amd -linfera -lpthread -Wl,-rpath,'$ORIGIN/../../infera_amd'
// run:   capi_columns_loop <model.onnx> <rows> <threads,...> [reps]      (INFERA_ZERO_COPY_RECT=0: the pulling kernel for every chunk)
#include <sys/mman.h>
#include <sys/resource.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "infera.h"
#include "infera_hip.h"
using namespace infera;

static double cpu_seconds() {
  rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  return ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6 + ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const uint64_t rows = std::strtoull(argv[2], nullptr, 10) / 2048 * 2048;
  const int reps = argc > 4 ? std::atoi(argv[4]) : 3;
  const uint32_t K = 128;
  if (infera_load_model("m", argv[1]) != 0) {
    std::printf("load: %s\n", infera_last_error());
    return 1;
  }
  const size_t bytes = size_t(rows) * K * 4;
  float *table = static_cast<float *>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  for (size_t i = 0; i < size_t(rows) * K; i++) table[i] = float(int(i * 2654435761u >> 8) & 0xffff) * (1.0f / 32768.0f) - 1.0f;
  const bool staged = std::getenv("LOOP_STAGED") != nullptr;
  if (!staged && infera_hip_register_host_memory(table, bytes) != 0) {
    std::printf("register: %s\n", infera_last_error());
    return 1;
  }
  const uint64_t nchunks = rows / 2048;
  for (const char *p = argv[3]; *p;) {
    const int threads = int(std::strtol(p, const_cast<char **>(&p), 10));
    if (*p == ',') p++;
    double best = 1e30, cpu = 0;
    for (int rep = 0; rep < reps + 1; rep++) {
      std::atomic<uint64_t> next{0};
      std::atomic<int> failed{0};
      const double c0 = cpu_seconds();
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < threads; t++)
        th.emplace_back([&] {
          std::vector<InferaColumn> cols(K);
          for (;;) {
            const uint64_t g = next.fetch_add(60);  // a row group's worth of chunks per claim
            if (g >= nchunks) break;
            for (uint64_t c = g; c < g + 60 && c < nchunks; c++) {
              for (uint32_t j = 0; j < K; j++) cols[j] = InferaColumn{table + size_t(j) * rows + c * 2048, nullptr, INFERA_COL_FLOAT, 0};
              InferaInferenceResult r = infera_predict_columns("m", cols.data(), K, 2048);
              if (r.status != 0) failed = 1;
              infera_free_result(r);
            }
          }
        });
      for (auto &x : th) x.join();
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (failed) {
        std::printf("predict failed: %s\n", infera_last_error());
        return 1;
      }
      if (rep > 0) {
        best = std::min(best, sec);
        cpu += cpu_seconds() - c0;
      }
    }
    std::printf("threads %2d: %6.1f M rows/s, %5.1f us of CPU per chunk (engine only: infera_predict_columns in a loop, %s)\n", threads, rows / best / 1e6,
                cpu / reps / nchunks * 1e6, staged ? "staged" : "registered table");
  }
  return 0;
}
