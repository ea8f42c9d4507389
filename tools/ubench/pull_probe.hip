// pull_probe.hip -- what a KERNEL reading registered host memory reaches over PCIe (the zero-copy gather's ceiling), by registration
// flags and load shape; the copy engine (hipMemcpyAsync from the same registered range) beside it.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pull_probe tools/ubench/pull_probe.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <cstdio>
#include <cstring>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int U, bool NT>
__global__ __launch_bounds__(256) void pull(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n4) {
  const size_t stride = size_t(gridDim.x) * 256;
  size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) dst[i + u * stride] = v[u];
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

int main() {
  const size_t bytes = 1ull << 30, n4 = bytes / 16;
  CK(hipSetDevice(0));
  float *dev;
  CK(hipMalloc(&dev, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  struct { const char *name; unsigned flags; } regs[] = {{"default", hipHostRegisterDefault}, {"portable|mapped", hipHostRegisterPortable | hipHostRegisterMapped},
                                                         {"coarse-grained", hipExtHostRegisterCoarseGrained}};
  for (auto &rg : regs) {
    char *host = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    std::memset(host, 1, bytes);
    if (hipHostRegister(host, bytes, rg.flags) != hipSuccess) { std::printf("%-16s register failed\n", rg.name); (void)hipGetLastError(); munmap(host, bytes); continue; }
    void *dptr = nullptr;
    CK(hipHostGetDevicePointer(&dptr, host, 0));
    auto run = [&](const char *what, auto launch) {
      launch();
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(a);
      for (int i = 0; i < 3; i++) launch();
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, a, b);
      std::printf("%-16s %-34s %6.1f GB/s\n", rg.name, what, bytes * 3.0 / ms / 1e6);
    };
    run("hipMemcpyAsync (copy engine)", [&] { (void)hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, 0); });
    for (int blocks : {128, 512, 2048}) {
      char w[64];
      std::snprintf(w, sizeof w, "kernel x1      %4d blocks", blocks);
      run(w, [&] { hipLaunchKernelGGL((pull<1, false>), dim3(blocks), dim3(256), 0, 0, (const f32x4 *)dptr, (f32x4 *)dev, n4); });
      std::snprintf(w, sizeof w, "kernel x4      %4d blocks", blocks);
      run(w, [&] { hipLaunchKernelGGL((pull<4, false>), dim3(blocks), dim3(256), 0, 0, (const f32x4 *)dptr, (f32x4 *)dev, n4); });
      std::snprintf(w, sizeof w, "kernel x4 nt   %4d blocks", blocks);
      run(w, [&] { hipLaunchKernelGGL((pull<4, true>), dim3(blocks), dim3(256), 0, 0, (const f32x4 *)dptr, (f32x4 *)dev, n4); });
      std::snprintf(w, sizeof w, "kernel x8      %4d blocks", blocks);
      run(w, [&] { hipLaunchKernelGGL((pull<8, false>), dim3(blocks), dim3(256), 0, 0, (const f32x4 *)dptr, (f32x4 *)dev, n4); });
    }
    CK(hipHostUnregister(host));
    munmap(host, bytes);
  }
  return 0;
}
