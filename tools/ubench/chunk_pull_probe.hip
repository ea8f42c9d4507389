// chunk_pull_probe.hip -- what caps the zero-copy fetch of ONE DataChunk-sized piece of a registered columnar host table (128 column runs of
// 8 KiB = 1 MiB) at ~43 GB/s when a kernel streaming a contiguous GiB reaches the copy engines' 57 (pull_probe.hip)?  Fetches of 1 MiB, issued
// back to back on S streams with nothing else in the queues: by mechanism (pull kernel shapes / hipMemcpy2DAsync / hipMemcpyAsync), by source
// pattern (128 strided runs vs one contiguous MiB) and by S.  If S >= 2 overlapped the link would fill; if they serialise the rate stays at one
// fetch's size over its latency.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/chunk_pull_probe tools/ubench/chunk_pull_probe.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kCols = 128, kRows = 2048;            // one chunk
constexpr size_t kColStride = size_t(4) << 20;      // floats between columns of the host table (16 MB): 128 columns = 2 GiB

// one workgroup per column run, 256 lanes x 2 loads of 16 bytes (the product's gather_cols_kernel for FLOAT columns)
__global__ __launch_bounds__(256) void pull_cols(const float *__restrict__ src, size_t col_stride, size_t r0, float *__restrict__ dst) {
  const f32x4 *s = reinterpret_cast<const f32x4 *>(src + size_t(blockIdx.x) * col_stride + r0);
  f32x4 *d = reinterpret_cast<f32x4 *>(dst + size_t(blockIdx.x) * kRows);
  const f32x4 a = s[threadIdx.x], b = s[threadIdx.x + 256];
  d[threadIdx.x] = a;
  d[threadIdx.x + 256] = b;
}
// one wave per column run: 64 lanes x 8 loads of 16 bytes, all issued before the first store (32 workgroups of 4 waves)
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
// 512 workgroups of one wave: one 1 KiB wave instruction... x2 (every run cut in four)
__global__ __launch_bounds__(64) void pull_cols_thin(const float *__restrict__ src, size_t col_stride, size_t r0, float *__restrict__ dst) {
  const int col = blockIdx.x >> 2, part = blockIdx.x & 3;
  const f32x4 *s = reinterpret_cast<const f32x4 *>(src + size_t(col) * col_stride + r0) + part * 128;
  f32x4 *d = reinterpret_cast<f32x4 *>(dst + size_t(col) * kRows) + part * 128;
  const f32x4 a = s[threadIdx.x], b = s[threadIdx.x + 64];
  d[threadIdx.x] = a;
  d[threadIdx.x + 64] = b;
}
__global__ void tiny(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }

int main(int argc, char **argv) {
  const int fetches = argc > 1 ? std::atoi(argv[1]) : 400;
  CK(hipSetDevice(0));
  const size_t bytes = kCols * kColStride * 4;
  float *host = (float *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  std::memset(host, 1, bytes);
  CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
  float *hdev = nullptr;
  CK(hipHostGetDevicePointer((void **)&hdev, host, 0));
  constexpr int kMaxS = 16;
  hipStream_t st[kMaxS];
  float *dev[kMaxS], *scratch;
  for (int i = 0; i < kMaxS; i++) {
    CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    CK(hipMalloc(&dev[i], size_t(kCols) * kRows * 4));
  }
  CK(hipMalloc(&scratch, 4096));
  CK(hipMemset(scratch, 0, 4096));
  struct Mech { const char *name; int id; };
  const Mech mechs[] = {{"pull kernel, 128 wg x 256 lanes x 2 loads", 0}, {"pull kernel, 32 wg, wave per run x 8 loads", 1}, {"pull kernel, 512 one-wave wg x 2 loads", 2},
                        {"hipMemcpy2DAsync", 3}, {"hipMemcpyAsync 1 MiB (contiguous only)", 4}};
  for (int follow = 0; follow < 2; follow++)
    for (int contiguous = 0; contiguous < 2; contiguous++)
      for (const Mech &m : mechs) {
        if (m.id == 4 && !contiguous) continue;
        std::printf("%-46s %-22s %s\n   ", m.name, contiguous ? "one contiguous MiB" : "128 runs of 8 KiB", follow ? "+ a small dependent kernel behind every fetch" : "");
        for (int S : {1, 2, 3, 4, 8, 16}) {
          // contiguous: columns 8 KiB apart (col_stride = 2048 floats) = 1 MiB in one piece; chunk c starts c MiB in
          const size_t col_stride = contiguous ? size_t(kRows) : kColStride;
          auto issue = [&](int s, int c) {
            const size_t r0 = contiguous ? size_t(c % 1024) * (size_t(kCols) * kRows) : size_t(c % 1024) * kRows;
            switch (m.id) {
              case 0: hipLaunchKernelGGL(pull_cols, dim3(kCols), dim3(256), 0, st[s], hdev, col_stride, r0, dev[s]); break;
              case 1: hipLaunchKernelGGL(pull_cols_wave, dim3(kCols / 4), dim3(256), 0, st[s], hdev, col_stride, r0, dev[s]); break;
              case 2: hipLaunchKernelGGL(pull_cols_thin, dim3(kCols * 4), dim3(64), 0, st[s], hdev, col_stride, r0, dev[s]); break;
              case 3: (void)hipMemcpy2DAsync(dev[s], size_t(kRows) * 4, host + r0, col_stride * 4, size_t(kRows) * 4, kCols, hipMemcpyHostToDevice, st[s]); break;
              default: (void)hipMemcpyAsync(dev[s], host + r0, size_t(kCols) * kRows * 4, hipMemcpyHostToDevice, st[s]); break;
            }
            if (follow) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, st[s], scratch);
          };
          for (int s = 0; s < S; s++) issue(s, s);
          (void)hipDeviceSynchronize();
          const auto t0 = std::chrono::steady_clock::now();
          for (int c = 0; c < fetches; c++) issue(c % S, c);
          (void)hipDeviceSynchronize();
          const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          std::printf(" S=%-2d %5.1f GB/s (%5.1f us)", S, double(fetches) * kCols * kRows * 4 / sec / 1e9, sec / fetches * 1e6);
        }
        std::printf("\n");
        if (hipGetLastError() != hipSuccess) std::printf("   (error)\n");
      }
  return 0;
}
