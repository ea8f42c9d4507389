// C++ counterpart of the reference's concurrency harness (test/concurrency/test_concurrency.py:25-50, 71-78)
// against the C ABI of include/infera.h -- plain C calls from std::threads, no Python in the process:
//   8 threads x 10 iterations: load `lin_{t}_{i}` from linear.onnx, predict (1,2,3) -> 1.75 +- 1e-5, unload;
//   an extra unload of a name that was never loaded fails with -1 and must not disturb anything;
//   finally infera_get_loaded_models() == "[]".
// usage: concurrency_harness <linear.onnx> [--no-predict]     (--no-predict: a box without a GPU -- the predict call must then
// fail with the loud "HIP backend unavailable" error, never compute; everything else is checked as above)
// exit code 0 = all checks passed; every failed check prints a line to stderr.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "infera.h"

using namespace infera;  // (the header wraps its extern "C" block in namespace infera for C++, as rust.h does)

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <linear.onnx> [--no-predict]\n", argv[0]);
    return 2;
  }
  const std::string path = argv[1];
  const bool no_predict = argc > 2 && std::strcmp(argv[2], "--no-predict") == 0;
  std::atomic<int> failures{0}, predictions{0};
  auto fail = [&](const std::string &what) {
    failures.fetch_add(1);
    const char *e = infera_last_error();
    std::fprintf(stderr, "FAIL: %s (last error on this thread: %s)\n", what.c_str(), e ? e : "<none>");
  };
  std::vector<std::thread> threads;
  for (int t = 0; t < 8; t++)
    threads.emplace_back([&, t] {
      for (int i = 0; i < 10; i++) {
        const std::string name = "lin_" + std::to_string(t) + "_" + std::to_string(i);
        if (infera_load_model(name.c_str(), path.c_str()) != 0) {
          fail("load " + name);
          continue;
        }
        const float x[3] = {1.f, 2.f, 3.f};
        InferaInferenceResult r = infera_predict(name.c_str(), x, 1, 3);
        if (no_predict) {
          const char *e = infera_last_error();
          if (r.status == 0 || r.data != nullptr || !e || !std::strstr(e, "HIP backend unavailable")) fail("predict on a box without a GPU must fail loudly: " + name);
        } else if (r.status != 0 || r.rows != 1 || r.cols != 1 || r.len != 1 || !r.data || std::fabs(r.data[0] - 1.75f) > 1e-5f) {
          fail("predict " + name);
        } else {
          predictions.fetch_add(1);
        }
        infera_free_result(r);  // (callers free on failure too: NULL data is a no-op)
        if (infera_unload_model(name.c_str()) != 0) fail("unload " + name);
      }
      if (infera_unload_model("non_existent_again") != -1) fail("unload of a missing model must return -1");
    });
  for (auto &th : threads) th.join();
  char *loaded = infera_get_loaded_models();
  if (!loaded || std::strcmp(loaded, "[]") != 0) fail(std::string("registry not empty: ") + (loaded ? loaded : "<null>"));
  infera_free(loaded);
  std::printf("{\"threads\": 8, \"iterations\": 10, \"predictions_ok\": %d, \"failures\": %d}\n", predictions.load(), failures.load());
  return failures.load() == 0 ? 0 : 1;
}
