// hostreg_probe.hip -- what pinning the CALLER's memory costs: hipHostRegister / hipHostUnregister of page-aligned ranges of a
// touched anonymous mapping, by range size (a DuckDB column segment of one row group is 480 KB; a 2048-row run 8 KB).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/hostreg_probe tools/ubench/hostreg_probe.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstring>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

int main() {
  const size_t total = 256ull << 20;
  char *base = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  std::memset(base, 1, total);
  CK(hipSetDevice(0));
  void *d = nullptr;
  CK(hipMalloc(&d, 1 << 20));
  for (size_t bytes : {size_t(8) << 10, size_t(64) << 10, size_t(480) << 10, size_t(4) << 20, size_t(61) << 20}) {
    const int n = int(std::min<size_t>(64, total / bytes));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) CK(hipHostRegister(base + size_t(i) * bytes, bytes, hipHostRegisterDefault));
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) CK(hipHostUnregister(base + size_t(i) * bytes));
    auto t2 = std::chrono::steady_clock::now();
    std::printf("%8zu KB x %2d: register %8.1f us each, unregister %8.1f us each\n", bytes >> 10, n,
                std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t1).count() / n);
  }
  return 0;
}
