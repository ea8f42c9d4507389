// hbm_stream.hip -- what a read-mostly stream can reach on this box (ceiling for the HBM-bound kernels).
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_stream hbm_stream.hip ; run: ./hbm_stream [GiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const f32x4 *__restrict__ x, float *__restrict__ out, size_t n4) {
  const size_t stride = size_t(gridDim.x) * 256;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += v[u];
  }
  for (; i < n4; i += stride) acc += x[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[0] = acc[0];
}

// read 128 floats, write 10 per row: the C4 traffic shape (every loaded value feeds a stored one)
__global__ __launch_bounds__(256) void read_write_kernel(const f32x4 *__restrict__ x, float *__restrict__ y, size_t rows) {
  const size_t stride = size_t(gridDim.x) * 256;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < rows * 32; i += stride) {
    const f32x4 v = x[i];
    float s = v[0] + v[1] + v[2] + v[3];
    s += __shfl_xor(s, 16);  // lanes 0..15 and 16..31 of a row (a row = 32 lanes = half a wave)
    if ((i & 31) < 10) y[(i >> 5) * 10 + (i & 31)] = s + __shfl_down(s, 10);
  }
}

// the same traffic with the reads issued the way a real kernel can: a wave owns 32 consecutive rows (16 KB), every lane
// has U 16-byte loads in flight, and the wave writes its 320 result floats as one contiguous run
template <int U>
__global__ __launch_bounds__(256) void read_write_tiled_kernel(const f32x4 *__restrict__ x, float *__restrict__ y, size_t rows) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t(blockIdx.x) * 256 + threadIdx.x) >> 6, nwaves = (size_t(gridDim.x) * 256) >> 6;
  for (size_t t = wave; t * 32 + 31 < rows; t += nwaves) {
    const f32x4 *src = x + t * 1024 + lane;  // 32 rows x 32 quads
    float s = 0.f;
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = src[(i0 + u) * 64];
#pragma unroll
      for (int u = 0; u < U; u++) s += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    }
    float *dst = y + t * 320;
#pragma unroll
    for (int j = 0; j < 5; j++) dst[j * 64 + lane] = s + float(j);
  }
}

__global__ __launch_bounds__(256) void copy_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, size_t n4) {
  const size_t stride = size_t(gridDim.x) * 256;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += stride) y[i] = x[i];
}

#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <typename F>
static double time_ms(F &&f, int reps = 10) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 24.0;
  const size_t bytes = size_t(gib * (1ull << 30)) / 512 * 512, n4 = bytes / 16, rows = bytes / 512;
  float *x, *y;
  CK(hipMalloc(&x, bytes));
  CK(hipMalloc(&y, bytes));
  CK(hipMemset(x, 0, bytes));
  for (int blocks : {256 * 2, 256 * 4, 256 * 8, 256 * 16}) {
    double t1 = time_ms([&] { hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, n4); });
    double t4 = time_ms([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, n4); });
    double t8 = time_ms([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, n4); });
    double tc = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, (f32x4 *)y, n4 / 2); });
    double tw = time_ms([&] { hipLaunchKernelGGL(read_write_kernel, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, rows); });
    double t8w = time_ms([&] { hipLaunchKernelGGL(read_write_tiled_kernel<8>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, rows); });
    double t16w = time_ms([&] { hipLaunchKernelGGL(read_write_tiled_kernel<16>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, rows); });
    printf("blocks %5d: tiled read128+write10, 8 / 16 loads in flight per lane: %.2f / %.2f TB/s\n", blocks, (bytes + rows * 40.0) / t8w / 1e9,
           (bytes + rows * 40.0) / t16w / 1e9);
    printf("blocks %5d: read x1 %.2f TB/s  x4 %.2f  x8 %.2f | copy %.2f TB/s (r+w) | read128+write10 %.2f TB/s\n", blocks, bytes / t1 / 1e9,
           bytes / t4 / 1e9, bytes / t8 / 1e9, bytes / tc / 1e9, (bytes + rows * 40.0) / tw / 1e9);
  }
  return 0;
}
