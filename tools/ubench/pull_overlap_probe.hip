// pull_overlap_probe.hip -- (round 6) why four callers of the registered scan reach 72-76 M rows/s where a closed queueing model around the link
// gives 86: per chunk a caller runs [pulling kernel: 1 MiB of 128 column runs out of registered host memory] -> [model kernel, ~8 us] -> waits.
// Here the same pair of kernels (the pull = the product's launch shape; the model kernel = 128 workgroups spinning for a given time, writing
// 8 KB into pinned host memory like the product's direct_out epilogue) on S streams, three ways:
//   free     S streams, every chunk enqueued ahead, no host wait at all          -> what the DEVICE can overlap
//   spin     S host threads, one chunk in flight each, hipEventQuery in a tight loop -> the synchronous ABI with zero wake latency
//   nap      the same with 2-us naps between queries (the product's wait)
// M rows/s (2048 rows per chunk).  build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pull_overlap_probe tools/ubench/pull_overlap_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/prctl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(_e)); std::exit(1); } } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kCols = 128, kRows = 2048;
constexpr size_t kColStride = size_t(4) << 20;  // floats between columns of the host table (16 MB)

__global__ __launch_bounds__(256) void pull_cols_wave(const float *__restrict__ src, size_t col_stride, size_t r0, float *__restrict__ dst) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const f32x4 *s = reinterpret_cast<const f32x4 *>(src + size_t(col) * col_stride + r0);
  f32x4 *d = reinterpret_cast<f32x4 *>(dst + size_t(col) * kRows);
  f32x4 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = s[lane + 64 * i];
#pragma unroll
  for (int i = 0; i < 8; i++) d[lane + 64 * i] = v[i];
}
// the model kernel's stand-in: 128 workgroups, each busy for `cycles` of the 100 MHz clock after reading its rows, 16 floats of result per workgroup
__global__ __launch_bounds__(256) void model_like(const float *__restrict__ x, float *__restrict__ y, unsigned cycles) {
  const float a = x[(blockIdx.x * 16 + (threadIdx.x & 15)) + size_t(threadIdx.x >> 4) * kRows];
  const unsigned long long t0 = __builtin_readcyclecounter();
  float acc = a;
  while (__builtin_readcyclecounter() - t0 < cycles) acc = acc * 1.0001f + 0.5f;
  if (threadIdx.x < 16) y[blockIdx.x * 16 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
  const int chunks = argc > 1 ? std::atoi(argv[1]) : 2000;
  const unsigned cycles = argc > 2 ? unsigned(std::atoi(argv[2])) : 700;  // ~7 us at 100 MHz
  CK(hipSetDevice(0));
  const size_t bytes = kCols * kColStride * 4;
  float *host = (float *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  std::memset(host, 1, bytes);
  CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
  float *hdev = nullptr;
  CK(hipHostGetDevicePointer((void **)&hdev, host, 0));
  constexpr int kMaxS = 16;
  hipStream_t st[kMaxS];
  float *dev[kMaxS], *pin[kMaxS];
  hipEvent_t ev[kMaxS];
  for (int i = 0; i < kMaxS; i++) {
    CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    CK(hipMalloc(&dev[i], size_t(kCols) * kRows * 4));
    CK(hipHostMalloc((void **)&pin[i], 8192, hipHostMallocDefault));
    CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  }
  const size_t nblocks = kColStride / kRows;  // row blocks available per column
  auto enqueue = [&](int s, size_t c) {
    hipLaunchKernelGGL(pull_cols_wave, dim3(32), dim3(256), 0, st[s], hdev, kColStride, ((c * 7919u + unsigned(s) * 131u) % nblocks) * kRows, dev[s]);
    hipLaunchKernelGGL(model_like, dim3(128), dim3(256), 0, st[s], dev[s], pin[s], cycles);
  };
  for (int s = 0; s < kMaxS; s++) enqueue(s, 0);
  CK(hipDeviceSynchronize());
  // single-kernel durations on an idle device
  {
    hipEvent_t a, b, c;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
    float tp = 0, tm = 0;
    for (int i = 0; i < 20; i++) {
      CK(hipEventRecord(a, st[0]));
      hipLaunchKernelGGL(pull_cols_wave, dim3(32), dim3(256), 0, st[0], hdev, kColStride, size_t(i) * kRows, dev[0]);
      CK(hipEventRecord(b, st[0]));
      hipLaunchKernelGGL(model_like, dim3(128), dim3(256), 0, st[0], dev[0], pin[0], cycles);
      CK(hipEventRecord(c, st[0]));
      CK(hipEventSynchronize(c));
      float x, y;
      CK(hipEventElapsedTime(&x, a, b)); CK(hipEventElapsedTime(&y, b, c));
      tp += x; tm += y;
    }
    std::printf("alone: pull %.1f us, model-like kernel %.1f us (events around each; includes ~launch gaps)\n", tp / 20 * 1e3, tm / 20 * 1e3);
  }
  for (int S : {1, 2, 3, 4, 6, 8, 16}) {
    // free-running
    auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < chunks; c++)
      for (int s = 0; s < S; s++) enqueue(s, size_t(c));
    CK(hipDeviceSynchronize());
    const double free_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double rate[2];
    for (int mode = 0; mode < 2; mode++) {
      std::atomic<int> ready{0};
      std::atomic<bool> go{false};
      std::vector<std::thread> th;
      for (int s = 0; s < S; s++)
        th.emplace_back([&, s] {
          (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
          (void)hipSetDevice(0);
          ready.fetch_add(1);
          while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
          for (int c = 0; c < chunks; c++) {
            enqueue(s, size_t(c));
            (void)hipEventRecord(ev[s], st[s]);
            while (hipEventQuery(ev[s]) == hipErrorNotReady)
              if (mode == 1) std::this_thread::sleep_for(std::chrono::microseconds(2));
          }
        });
      while (ready.load() < S) std::this_thread::yield();
      t0 = std::chrono::steady_clock::now();
      go.store(true, std::memory_order_release);
      for (auto &x : th) x.join();
      rate[mode] = double(S) * chunks * kRows / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e6;
    }
    std::printf("S = %2d: free-running %6.1f M rows/s (%5.1f GB/s of pulls) | one chunk in flight per thread: spin %6.1f | 2-us naps %6.1f M rows/s\n", S,
                double(S) * chunks * kRows / free_s / 1e6, double(S) * chunks * (1 << 20) / free_s / 1e9, rate[0], rate[1]);
  }
  return 0;
}
