// H2D ceiling by transfer size: T threads, each looping {hipMemcpyAsync(size) ; wait} on its own stream from its own
// pinned buffer; and the same bytes pulled by a kernel straight from pinned memory (zero-copy) with G blocks.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/h2d_size tools/ubench/h2d_size.hip -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); std::exit(1); } } while (0)

__global__ void pull_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i + 3 * stride < n; i += 4 * stride) {  // four independent 16-byte loads in flight per lane
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

int main() {
  const size_t total = size_t(4) << 30;  // bytes moved per measurement
  for (int pull = 0; pull < 2; pull++)
    for (size_t mib : {1, 2, 4, 8, 16}) {
      for (int T : {2, 4, 8}) {
        const size_t bytes = mib << 20, iters = total / bytes / T;
        auto worker = [&] {
          hipStream_t s;
          CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
          char *pin, *dev;
          CK(hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault));
          CK(hipMalloc((void **)&dev, bytes));
          std::memset(pin, 1, bytes);
          hipEvent_t ev;
          CK(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
          for (size_t i = 0; i < iters; i++) {
            if (pull) hipLaunchKernelGGL(pull_kernel, dim3(unsigned(std::min<size_t>(256, bytes / 16 / 256 / 4))), dim3(256), 0, s, (const float4 *)pin, (float4 *)dev, bytes / 16);
            else CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev, s));
            CK(hipEventSynchronize(ev));
          }
          CK(hipFree(dev));
          CK(hipHostFree(pin));
          CK(hipStreamDestroy(s));
        };
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(worker);
        for (auto &x : th) x.join();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%s size=%2zu MiB threads=%d  %6.1f GB/s\n", pull ? "kernel-pull  " : "hipMemcpyAsync", mib, T, double(iters) * T * bytes / sec / 1e9);
        std::fflush(stdout);
      }
    }
  return 0;
}
