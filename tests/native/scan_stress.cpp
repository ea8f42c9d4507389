// Stress harness for the host path's shared state: plain C calls from std::threads on the C ABI, no Python in the
// process.  16 scanner threads call infera_predict_columns on 2048-row chunks while
//   * 2 threads register / unregister 256 KB blocks that the scanners are READING columns from (a database allocator's hook would do this:
//     allocate -> scan -> free while other threads scan) -- a chunk whose blocks are all registered is read in place by the GPU, any other
//     chunk takes the staged path; either way the result must be bit for bit the expected one;
//   * 2 threads load a model, predict (1,2,3) -> 1.75, unload it (the reference's concurrency contract, test_concurrency.py:25-50).
// Phase A runs the scanners alone, phase B with the four disturbers: the JSON line reports both rates, so a test can bound what the churn
// costs the scan.  Built twice: plain (tests/test_native_harness.py) and under ThreadSanitizer against a TSan build of the host objects
// (`make -C tests/native tsan`; profiles/r04_tsan.txt).
// usage: scan_stress <mlp128.onnx> <linear.onnx> [seconds per phase = 2] [scanner threads = 16]
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "infera_hip.h"

using namespace infera;

namespace {
constexpr int K = 128, CHUNK = 2048, NCHUNKS = 64, CHURN_CHUNKS = 16, BLOCK_COLS = 32, BLOCKS_PER_CHUNK = K / BLOCK_COLS;
constexpr size_t BLOCK_BYTES = size_t(BLOCK_COLS) * CHUNK * 4;  // 256 KB

std::atomic<int> g_failures{0};
void fail(const std::string &what) {
  g_failures.fetch_add(1);
  const char *e = infera_last_error();
  std::fprintf(stderr, "FAIL: %s (last error on this thread: %s)\n", what.c_str(), e ? e : "<none>");
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <mlp128.onnx> <linear.onnx> [seconds] [threads]\n", argv[0]);
    return 2;
  }
  const double seconds = argc > 3 ? std::atof(argv[3]) : 2.0;
  const int nscan = argc > 4 ? std::atoi(argv[4]) : 16;
  if (infera_hip_device_count() <= 0) {  // without a GPU every predict fails loudly: nothing to scan (the plain concurrency harness covers that box)
    std::printf("{\"gpu\": false, \"failures\": 0}\n");
    return 0;
  }
  if (infera_load_model("stress_mlp", argv[1]) != 0) {
    fail("load stress_mlp");
    return 1;
  }
  // the table: [NCHUNKS] row groups of CHUNK rows, one contiguous run per column inside a group (a columnar store's layout)
  const size_t table_floats = size_t(NCHUNKS) * K * CHUNK;
  float *table = static_cast<float *>(std::aligned_alloc(4096, table_floats * 4));
  uint64_t s = 42;
  for (size_t i = 0; i < table_floats; i++) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    table[i] = float(int32_t(s >> 40) - (1 << 23)) * (1.0f / float(1 << 23));
  }
  auto table_cols = [&](int chunk, InferaColumn *cols) {
    for (int c = 0; c < K; c++) cols[c] = InferaColumn{table + (size_t(chunk) * K + size_t(c)) * CHUNK, nullptr, INFERA_COL_FLOAT, 0};
  };
  // expected results: every chunk once through the STAGED path (nothing is registered yet)
  std::vector<float> expected(size_t(NCHUNKS) * CHUNK);
  for (int i = 0; i < NCHUNKS; i++) {
    InferaColumn cols[K];
    table_cols(i, cols);
    InferaInferenceResult r = infera_predict_columns("stress_mlp", cols, K, CHUNK);
    if (r.status != 0 || r.len != size_t(CHUNK)) {
      fail("reference chunk " + std::to_string(i));
      infera_free_result(r);
      return 1;
    }
    std::memcpy(&expected[size_t(i) * CHUNK], r.data, size_t(CHUNK) * 4);
    infera_free_result(r);
  }
  // churn blocks: churn chunk j = a copy of table chunk j in four separately allocated 256 KB blocks of 32 column runs each
  std::vector<float *> blocks(size_t(CHURN_CHUNKS) * BLOCKS_PER_CHUNK);
  for (int j = 0; j < CHURN_CHUNKS; j++)
    for (int b = 0; b < BLOCKS_PER_CHUNK; b++) {
      float *blk = static_cast<float *>(std::aligned_alloc(4096, BLOCK_BYTES));
      std::memcpy(blk, table + (size_t(j) * K + size_t(b) * BLOCK_COLS) * CHUNK, BLOCK_BYTES);
      blocks[size_t(j) * BLOCKS_PER_CHUNK + size_t(b)] = blk;
    }
  auto churn_cols = [&](int j, InferaColumn *cols) {
    for (int c = 0; c < K; c++)
      cols[c] = InferaColumn{blocks[size_t(j) * BLOCKS_PER_CHUNK + size_t(c / BLOCK_COLS)] + size_t(c % BLOCK_COLS) * CHUNK, nullptr, INFERA_COL_FLOAT, 0};
  };
  const bool gpu = true;
  if (gpu && infera_hip_register_host_memory(table, table_floats * 4) != 0) fail("register table");
  for (float *blk : blocks)
    if (gpu && infera_hip_register_host_memory(blk, BLOCK_BYTES) != 0) fail("register block");

  auto run_phase = [&](bool disturb, uint64_t &chunks_done, uint64_t &churn_ops, uint64_t &model_ops) {
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> done{0}, cops{0}, mops{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nscan; t++)
      th.emplace_back([&, t] {
        InferaColumn cols[K];
        for (uint64_t i = uint64_t(t); !stop.load(std::memory_order_relaxed); i += uint64_t(nscan)) {
          const bool from_blocks = (i * 2654435761ull >> 7) % 4 == 0;  // a quarter of the chunks read the churned blocks
          const int chunk = from_blocks ? int(i % CHURN_CHUNKS) : int(i % NCHUNKS);
          from_blocks ? churn_cols(chunk, cols) : table_cols(chunk, cols);
          InferaInferenceResult r = infera_predict_columns("stress_mlp", cols, K, CHUNK);
          if (r.status != 0 || r.len != size_t(CHUNK) || std::memcmp(r.data, &expected[size_t(chunk) * CHUNK], size_t(CHUNK) * 4) != 0)
            fail("scan chunk " + std::to_string(chunk) + (from_blocks ? " (churned blocks)" : " (table)"));
          infera_free_result(r);
          done.fetch_add(1, std::memory_order_relaxed);
        }
      });
    if (disturb) {
      for (int t = 0; t < 2; t++)  // allocator-hook churn: each thread owns half of the blocks
        th.emplace_back([&, t] {
          for (size_t i = size_t(t); !stop.load(std::memory_order_relaxed); i += 2) {
            float *blk = blocks[i % blocks.size()];
            if (!gpu) {
              std::this_thread::sleep_for(std::chrono::microseconds(50));
              continue;
            }
            if (infera_hip_unregister_host_memory(blk) != 0) fail("unregister block");
            if (infera_hip_register_host_memory(blk, BLOCK_BYTES) != 0) fail("re-register block");
            cops.fetch_add(2, std::memory_order_relaxed);
          }
        });
      for (int t = 0; t < 2; t++)  // the reference's load / predict / unload loop
        th.emplace_back([&, t] {
          for (int i = 0; !stop.load(std::memory_order_relaxed); i++) {
            const std::string name = "lin_" + std::to_string(t) + "_" + std::to_string(i);
            if (infera_load_model(name.c_str(), argv[2]) != 0) {
              fail("load " + name);
              continue;
            }
            const float x[3] = {1.f, 2.f, 3.f};
            InferaInferenceResult r = infera_predict(name.c_str(), x, 1, 3);
            if (gpu && (r.status != 0 || r.len != 1 || std::fabs(r.data[0] - 1.75f) > 1e-5f)) fail("predict " + name);
            infera_free_result(r);
            if (infera_unload_model(name.c_str()) != 0) fail("unload " + name);
            mops.fetch_add(1, std::memory_order_relaxed);
          }
        });
    }
    const double t0 = now_s();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for (auto &x : th) x.join();
    chunks_done = done.load();
    churn_ops = cops.load();
    model_ops = mops.load();
    return now_s() - t0;
  };
  uint64_t warm, a_chunks, b_chunks, churn, models, z;
  run_phase(false, warm, z, z);  // contexts, staging, code objects
  const uint64_t zc0 = infera_hip_zero_copy_calls();
  const double ta = run_phase(false, a_chunks, z, z);
  const uint64_t zc1 = infera_hip_zero_copy_calls();
  const double tb = run_phase(true, b_chunks, churn, models);
  const uint64_t zc2 = infera_hip_zero_copy_calls();
  for (float *blk : blocks) (void)infera_hip_unregister_host_memory(blk);
  if (infera_hip_unregister_host_memory(table) != 0) fail("unregister table");
  if (infera_unload_model("stress_mlp") != 0) fail("unload stress_mlp");
  char *loaded = infera_get_loaded_models();
  if (!loaded || std::strcmp(loaded, "[]") != 0) fail(std::string("registry not empty: ") + (loaded ? loaded : "<null>"));
  infera_free(loaded);
  std::printf("{\"gpu\": true, \"scanners\": %d, \"quiet_rows_per_s\": %.0f, \"disturbed_rows_per_s\": %.0f, \"quiet_zero_copy_share\": %.3f, "
              "\"disturbed_zero_copy_share\": %.3f, \"register_unregister_ops\": %llu, \"load_predict_unload_ops\": %llu, \"failures\": %d}\n",
              nscan, double(a_chunks) * CHUNK / ta, double(b_chunks) * CHUNK / tb, double(zc1 - zc0) / double(a_chunks ? a_chunks : 1),
              double(zc2 - zc1) / double(b_chunks ? b_chunks : 1), (unsigned long long)churn, (unsigned long long)models, g_failures.load());
  return g_failures.load() == 0 ? 0 : 1;
}
