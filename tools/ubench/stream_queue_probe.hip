// stream_queue_probe.hip -- which hardware (HSA) queue does the runtime give the n-th stream of a process?  Creates streams one after the other
// (optionally each from its own thread), launches one marker kernel per stream with the stream's index as grid size, and leaves the mapping to
// `rocprofv3 --kernel-trace` (queue_id per dispatch).  Also times two spinning kernels on every PAIR of the first 8 streams: pairs that share a queue
// take twice as long.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/stream_queue_probe tools/ubench/stream_queue_probe.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void marker(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1; }
__global__ void spin(long long cycles) {
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 16;
  const bool threaded = argc > 2 && std::atoi(argv[2]) != 0;
  (void)hipSetDevice(0);
  int *d;
  (void)hipMalloc(&d, 4096);
  std::vector<hipStream_t> st(size_t(n), nullptr);
  for (int i = 0; i < n; i++) {
    auto make = [&, i] {
      (void)hipSetDevice(0);
      (void)hipStreamCreateWithFlags(&st[size_t(i)], hipStreamNonBlocking);
      hipLaunchKernelGGL(marker, dim3(unsigned(i + 1)), dim3(64), 0, st[size_t(i)], d);
      (void)hipStreamSynchronize(st[size_t(i)]);
    };
    if (threaded) std::thread(make).join();
    else make();
  }
  const int m = n < 8 ? n : 8;
  std::printf("pair time / single time (2.0 = the two streams share a hardware queue)\n     ");
  for (int b = 0; b < m; b++) std::printf("  s%-3d", b);
  std::printf("\n");
  auto run = [&](int a, int b) {
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[size_t(a)], 200000LL);
    if (b >= 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[size_t(b)], 200000LL);
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  (void)run(0, -1);
  const double single = run(0, -1);
  for (int a = 0; a < m; a++) {
    std::printf("s%-3d ", a);
    for (int b = 0; b < m; b++) std::printf(" %5.2f", a == b ? 0.0 : run(a, b) / single);
    std::printf("\n");
  }
  return 0;
}
