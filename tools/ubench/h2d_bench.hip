// What bounds the host path?  T threads, each with its own stream and a 1 MiB pinned staging buffer, doing per
// 2048-row x 128-col chunk: (g) the gather memcpy of 128 8-KiB column runs out of a multi-GB host table into pinned memory,
// (h) hipMemcpyAsync H2D of the chunk + wait, or both (gh) -- the host path minus the model.  Also (z): no H2D at all, a
// kernel reads the pinned chunk over PCIe itself (zero-copy) and writes it to HBM.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/h2d_bench tools/ubench/h2d_bench.hip -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); std::exit(1); } } while (0)

__global__ void pull_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) dst[i] = src[i];
}

int main(int argc, char **argv) {
  const size_t rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 10000000;
  const int ncols = 128, CH = 2048;
  const size_t RG = 122880;
  std::vector<float> table(rows * ncols);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < 16; t++) th.emplace_back([&, t] { for (size_t i = t; i < table.size(); i += 16 * 1024) std::memset(&table[i], 1, std::min<size_t>(1024, table.size() - i) * 4); });
    for (auto &x : th) x.join();
  }
  const size_t nchunks = rows / CH;
  for (const char *mode : {"g", "h", "gh", "z", "gz"}) {
    for (int T : {4, 8, 12, 16, 24, 32}) {
      std::atomic<size_t> next{0};
      auto worker = [&] {
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        float *pin, *dev;
        CK(hipHostMalloc((void **)&pin, size_t(CH) * ncols * 4, hipHostMallocDefault));
        CK(hipMalloc((void **)&dev, size_t(CH) * ncols * 4));
        const bool g = std::strchr(mode, 'g'), h = std::strchr(mode, 'h'), z = std::strchr(mode, 'z');
        for (;;) {
          const size_t c = next.fetch_add(1);
          if (c >= nchunks) break;
          if (g) {
            const size_t row0 = c * CH, g0 = row0 / RG * RG, gr = std::min(RG, rows - g0);
            const float *base = table.data() + g0 * ncols + (row0 - g0);
            for (int j = 0; j < ncols; j++) std::memcpy(pin + size_t(j) * CH, base + size_t(j) * gr, CH * 4);
          }
          if (h) {
            CK(hipMemcpyAsync(dev, pin, size_t(CH) * ncols * 4, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
          }
          if (z) {
            hipLaunchKernelGGL(pull_kernel, dim3(64), dim3(256), 0, s, (const float4 *)pin, (float4 *)dev, size_t(CH) * ncols / 4);
            CK(hipStreamSynchronize(s));
          }
        }
        CK(hipFree(dev));
        CK(hipHostFree(pin));
        CK(hipStreamDestroy(s));
      };
      {  // warm-up of contexts is inside (short); run twice, report the second
        double best = 0;
        for (int rep = 0; rep < 2; rep++) {
          next = 0;
          const auto t0 = std::chrono::steady_clock::now();
          std::vector<std::thread> th;
          for (int t = 0; t < T; t++) th.emplace_back(worker);
          for (auto &x : th) x.join();
          const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          best = double(nchunks) * CH * ncols * 4 / sec / 1e9;
        }
        std::printf("mode=%-2s threads=%2d  %6.1f GB/s  (%.1f M rows/s)\n", mode, T, best, best * 1e9 / 512 / 1e6);
        std::fflush(stdout);
      }
    }
  }
  return 0;
}
