// gather_bench.cpp -- host-only throughput of infera_gather_columns (the ExtractFeatures replacement) per
// column type, DuckDB-shaped: 2048-row chunks of a 128-column table into one reused staging buffer.
// build: g++ -O2 -std=c++17 -I include -o gather_bench tools/ubench/gather_bench.cpp -L infera_amd -linfera -Wl,-rpath,$PWD/infera_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "infera.h"
#include "infera_hip.h"

template <typename T>
static double run(int type, size_t rows, size_t ncols) {
  std::vector<std::vector<T>> data(ncols, std::vector<T>(rows));
  for (size_t c = 0; c < ncols; c++)
    for (size_t r = 0; r < rows; r++) data[c][r] = T((r * 131 + c * 17) % 1000) / T(7);
  std::vector<infera::InferaColumn> cols(ncols);
  for (size_t c = 0; c < ncols; c++) cols[c] = {data[c].data(), nullptr, type, 0};
  std::vector<float> out(2048 * ncols);
  double best = 1e30;
  for (int rep = 0; rep < 5; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (size_t r0 = 0; r0 + 2048 <= rows; r0 += 2048)
      if (infera::infera_gather_columns(cols.data(), ncols, r0, 2048, out.data()) != 0) exit(1);
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  return rows / best / 1e6;
}

int main() {
  const size_t rows = 2048 * 128, ncols = 128;
  printf("FLOAT   %.1f M rows/s\n", run<float>(infera::INFERA_COL_FLOAT, rows, ncols));
  printf("DOUBLE  %.1f M rows/s\n", run<double>(infera::INFERA_COL_DOUBLE, rows, ncols));
  printf("INTEGER %.1f M rows/s\n", run<int32_t>(infera::INFERA_COL_INTEGER, rows, ncols));
  printf("BIGINT  %.1f M rows/s\n", run<int64_t>(infera::INFERA_COL_BIGINT, rows, ncols));
  return 0;
}
