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

// (round 6) the tiled stream with the knobs the C4 kernel could still turn: results leaving as 16-byte pieces (80 per 32-row tile: 64 lanes + 16
// lanes) instead of five 4-byte-per-lane runs; non-temporal loads (NT & 1) / stores (NT & 2); CHUNKED: a wave walks a contiguous range of the
// table instead of striding over it by the grid
template <int U, int NT, bool CHUNKED>
__global__ __launch_bounds__(256) void read_write_tiled16_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, size_t rows) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t(blockIdx.x) * 256 + threadIdx.x) >> 6, nwaves = (size_t(gridDim.x) * 256) >> 6;
  const size_t ntiles = rows / 32, per = (ntiles + nwaves - 1) / nwaves;
  size_t t = CHUNKED ? wave * per : wave;
  const size_t tend = CHUNKED ? (t + per < ntiles ? t + per : ntiles) : ntiles, step = CHUNKED ? 1 : nwaves;
  for (; t < tend; t += step) {
    const f32x4 *src = x + t * 1024 + lane;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = (NT & 1) ? __builtin_nontemporal_load(src + (i0 + u) * 64) : src[(i0 + u) * 64];
#pragma unroll
      for (int u = 0; u < U; u++) s += v[u];
    }
    f32x4 *dst = y + t * 80;
    if (NT & 2) {
      __builtin_nontemporal_store(s, dst + lane);
      if (lane < 16) __builtin_nontemporal_store(s, dst + 64 + lane);
    } else {
      dst[lane] = s;
      if (lane < 16) dst[64 + lane] = s;
    }
  }
}

// store flavours by cache policy bits (MI355X_MICROARCH.md: plain / nt keep the line in the XCD's L2, sc1 / sc0 sc1 write through and drop it);
// ST: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt; LD: 0 plain, 1 nt; BURST tiles of results leave together (5 KB runs at BURST = 4)
template <int ST>
__device__ __forceinline__ void store16(f32x4 *p, f32x4 v) {
  if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (ST == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int LD, int ST, int BURST>
__global__ __launch_bounds__(256) void read_write_flavour_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, size_t rows) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t(blockIdx.x) * 256 + threadIdx.x) >> 6, nwaves = (size_t(gridDim.x) * 256) >> 6;
  const size_t ngroups = rows / 32 / BURST;
  for (size_t g = wave; g < ngroups; g += nwaves) {
    f32x4 s[BURST];
#pragma unroll
    for (int b = 0; b < BURST; b++) {
      const f32x4 *src = x + (g * BURST + b) * 1024 + lane;
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (LD == 0) v[u] = src[u * 64];
        if (LD == 1) v[u] = __builtin_nontemporal_load(src + u * 64);
      }
      s[b] = v[0];
#pragma unroll
      for (int u = 1; u < 16; u++) s[b] += v[u];
    }
    f32x4 *dst = y + g * BURST * 80;
#pragma unroll
    for (int b = 0; b < BURST; b++) {
      store16<ST>(dst + b * 80 + lane, s[b]);
      if (lane < 16) store16<ST>(dst + b * 80 + 64 + lane, s[b]);
    }
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
  for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
    double t1 = time_ms([&] { hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, n4); });
    double t4 = time_ms([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, n4); });
    double t8 = time_ms([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, n4); });
    double tc = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, (f32x4 *)y, n4 / 2); });
    double tw = time_ms([&] { hipLaunchKernelGGL(read_write_kernel, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, rows); });
    double t8w = time_ms([&] { hipLaunchKernelGGL(read_write_tiled_kernel<8>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, rows); });
    double t16w = time_ms([&] { hipLaunchKernelGGL(read_write_tiled_kernel<16>, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, y, rows); });
    printf("blocks %5d: tiled read128+write10, 8 / 16 loads in flight per lane: %.2f / %.2f TB/s\n", blocks, (bytes + rows * 40.0) / t8w / 1e9,
           (bytes + rows * 40.0) / t16w / 1e9);
    {
      auto rate = [&](double t) { return (bytes + rows * 40.0) / t / 1e9; };
#define RW16(U, NT, CH) rate(time_ms([&] { hipLaunchKernelGGL((read_write_tiled16_kernel<U, NT, CH>), dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, (f32x4 *)y, rows); }))
      printf("blocks %5d: tiled, 16-byte result pieces, 16 loads in flight: plain %.2f | nt loads %.2f | nt stores %.2f | both %.2f | chunked %.2f | chunked nt both %.2f | 8 in flight %.2f TB/s\n",
             blocks, RW16(16, 0, false), RW16(16, 1, false), RW16(16, 2, false), RW16(16, 3, false), RW16(16, 0, true), RW16(16, 3, true), RW16(8, 0, false));
    }
    {
      auto rate = [&](double t) { return (bytes + rows * 40.0) / t / 1e9; };
#define FL(LD, ST, B) rate(time_ms([&] { hipLaunchKernelGGL((read_write_flavour_kernel<LD, ST, B>), dim3(blocks), dim3(256), 0, 0, (const f32x4 *)x, (f32x4 *)y, rows); }))
      printf("blocks %5d: store flavours (plain loads): plain %.2f | nt %.2f | sc1 %.2f | sc0 sc1 %.2f | sc1 nt %.2f || nt loads: plain %.2f | nt %.2f | sc1 %.2f | sc0 sc1 %.2f | sc1 nt %.2f TB/s\n", blocks,
             FL(0, 0, 1), FL(0, 1, 1), FL(0, 2, 1), FL(0, 3, 1), FL(0, 4, 1), FL(1, 0, 1), FL(1, 1, 1), FL(1, 2, 1), FL(1, 3, 1), FL(1, 4, 1));
      printf("blocks %5d: bursts of 4 tiles: plain/plain %.2f | nt/nt %.2f | nt/sc1 %.2f TB/s\n", blocks, FL(0, 0, 4), FL(1, 1, 4), FL(1, 2, 4));
    }
    printf("blocks %5d: read x1 %.2f TB/s  x4 %.2f  x8 %.2f | copy %.2f TB/s (r+w) | read128+write10 %.2f TB/s\n", blocks, bytes / t1 / 1e9,
           bytes / t4 / 1e9, bytes / t8 / 1e9, bytes / tc / 1e9, (bytes + rows * 40.0) / tw / 1e9);
  }
  return 0;
}
