// mfma_order_probe.hip -- do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 accumulate a K-long dot product in the same order, bit for bit, and is
// that order the k-ascending fmaf chain?  (a 16-row form of the per-chunk MLP kernel would have to reproduce the scan kernel's sums.)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/mfma_order_probe tools/ubench/mfma_order_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int K = 256;

__global__ void k32(const float *A, const float *B, float *D) {  // A[32][K], B[K][32] -> D[32][32]
  const int l = threadIdx.x, n = l & 31, h = l >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + k + h], B[(k + h) * 32 + n], acc, 0, 0, 0);
  for (int i = 0; i < 16; i++) D[((i / 4) * 8 + h * 4 + (i % 4)) * 32 + n] = acc[i];
}
__global__ void k16(const float *A, const float *B, float *D) {  // the top-left 16 x 16 block of the same product
  const int l = threadIdx.x, n = l & 15, q = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * K + k + q], B[(k + q) * 32 + n], acc, 0, 0, 0);
  for (int i = 0; i < 4; i++) D[(q * 4 + i) * 16 + n] = acc[i];
}

int main() {
  std::vector<float> A(32 * K), B(K * 32), D32(32 * 32), D16(16 * 16);
  srand(7);
  for (auto &x : A) x = float(rand()) / RAND_MAX * 2.f - 1.f;
  for (auto &x : B) x = (float(rand()) / RAND_MAX * 2.f - 1.f) * 0.37f;
  float *dA, *dB, *dD32, *dD16;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dD32, D32.size() * 4); (void)hipMalloc(&dD16, D16.size() * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD32);
  hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD16);
  (void)hipMemcpy(D32.data(), dD32, D32.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(D16.data(), dD16, D16.size() * 4, hipMemcpyDeviceToHost);
  int d_32_chain = 0, d_16_chain = 0, d_32_16 = 0, d_16_pair = 0;
  for (int m = 0; m < 16; m++)
    for (int n = 0; n < 16; n++) {
      float c = 0.f, p = 0.f;
      for (int k = 0; k < K; k++) c = std::fmaf(A[m * K + k], B[k * 32 + n], c);
      for (int k = 0; k < K; k += 4) {  // alternative: four products summed pairwise, then added
        const float s01 = std::fmaf(A[m * K + k + 1], B[(k + 1) * 32 + n], A[m * K + k] * B[k * 32 + n]);
        const float s23 = std::fmaf(A[m * K + k + 3], B[(k + 3) * 32 + n], A[m * K + k + 2] * B[(k + 2) * 32 + n]);
        p += s01 + s23;
      }
      d_32_chain += std::memcmp(&D32[m * 32 + n], &c, 4) != 0;
      d_16_chain += std::memcmp(&D16[m * 16 + n], &c, 4) != 0;
      d_32_16 += std::memcmp(&D32[m * 32 + n], &D16[m * 16 + n], 4) != 0;
      d_16_pair += std::memcmp(&D16[m * 16 + n], &p, 4) != 0;
    }
  std::printf("K = %d, 256 dot products: 32x32x2 vs k-ascending fmaf chain: %d differ; 16x16x4 vs the chain: %d differ; 32x32x2 vs 16x16x4: %d differ; (16x16x4 vs pairwise-4: %d differ)\n",
              K, d_32_chain, d_16_chain, d_32_16, d_16_pair);
  return 0;
}
