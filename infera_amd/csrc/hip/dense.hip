// dense.hip -- generic fused dense layer  Y[rows,M] = act(X[rows,K] . W[K,M] + bias)  on the
// exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32), any K / M / rows.
//
// Formulation (wave64, transposed so activations sit on the MFMA N axis):
//     D^T[m, r] += sum_k  W^T[m, k] * X^T[k, r]        A-operand = W (lane: m = lane&31, k = lane>>5)
//                                                       B-operand = X (lane: r = lane&31, k = lane>>5)
// A wave owns 32 table rows and MT 32-wide output tiles; its X rows are staged through LDS in
// 64-column chunks with coalesced 16 B loads (row stride 68 floats: ds_write_b128 / ds_read_b128
// both conflict-free, see MI355X_MICROARCH.md LDS table), W fragments come straight from L1/L2
// (W is small and shared by every wave).  One ds_read_b128 feeds four MFMA k-steps: lane half h
// supplies k = 8g + 4h + j for step j, so the k order inside each group of 8 is 0,4,1,5,2,6,3,7 --
// a fixed permutation of the fp32 summation order, nothing else.
// Epilogue (bias, activation, optional row softmax) runs on the accumulator registers.
//
// Bound: MFMA for wide layers (2*K*M flop/row vs 4*(K+M) B/row), HBM for narrow ones such as
// the C4 logistic regression (128 -> 10: 2,560 flop/row vs 552 B/row).
#include "device_common.hpp"

namespace infera_hip::kern {

namespace {

constexpr int KC = 64;         // K columns staged per chunk
constexpr int LDS_STRIDE = 68; // floats per staged row (16 B aligned, conflict-free for b128)
constexpr int WAVES = 4;

template <int MT, int SM>
__global__ __launch_bounds__(WAVES * 64) void dense_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                          const float *__restrict__ bias, float *__restrict__ Y,
                                                          int64_t rows, int K, int M, ActParam act, int vec_ok) {
  __shared__ __attribute__((aligned(16))) float xs[WAVES][32][LDS_STRIDE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t(blockIdx.x) * WAVES + wave) * 32;
  float(*tile)[LDS_STRIDE] = xs[wave];

  for (int m0 = 0; m0 < M; m0 += 32 * MT) {
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

    for (int k0 = 0; k0 < K; k0 += KC) {
      __syncthreads();  // previous chunk's reads are done
      if (vec_ok) {
        // 32 rows x 16 float4 = 512 slots, 8 per lane; a wave instruction covers 4 full 256 B rows
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int slot = i * 64 + lane, rr = slot >> 4, c4 = (slot & 15) * 4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          const int64_t gr = row0 + rr;
          if (gr < rows && k0 + c4 < K) v = *reinterpret_cast<const f32x4 *>(X + gr * K + k0 + c4);
          *reinterpret_cast<f32x4 *>(&tile[rr][c4]) = v;
        }
      } else {
        for (int idx = lane; idx < 32 * KC; idx += 64) {
          const int rr = idx >> 6, cc = idx & 63;
          const int64_t gr = row0 + rr;
          tile[rr][cc] = (gr < rows && k0 + cc < K) ? X[gr * K + k0 + cc] : 0.f;
        }
      }
      __syncthreads();
      const int kchunk = (K - k0) < KC ? (K - k0) : KC;
      for (int g = 0; g * 8 < kchunk; g++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(&tile[r][8 * g + 4 * h]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int k = k0 + 8 * g + 4 * h + j;
#pragma unroll
          for (int t = 0; t < MT; t++) {
            const int m = m0 + 32 * t + r;
            const float a = (k < K && m < M) ? W[int64_t(k) * M + m] : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b4[j], acc[t], 0, 0, 0);
          }
        }
      }
    }

    // ---- epilogue: lane (r,h) holds, for table row row0+r, features m0 + 32t + 8*(i>>2) + 4h + (i&3)
    const int64_t grow = row0 + r;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int f = m0 + 32 * t + 8 * (i >> 2) + 4 * h + (i & 3);
        float v = acc[t][i];
        if (bias != nullptr && f < M) v += bias[f];
        acc[t][i] = apply_act(v, act);
      }
    if constexpr (SM != 0) {  // row softmax over all M outputs (host guarantees M <= 32*MT)
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int f = 32 * t + 8 * (i >> 2) + 4 * h + (i & 3);
          if (f < M) mx = fmaxf(mx, acc[t][i]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int f = 32 * t + 8 * (i >> 2) + 4 * h + (i & 3);
          if (f < M) {
            const float e = expf(acc[t][i] - mx);
            sum += e;
            if (SM == 1) acc[t][i] = e;
            else acc[t][i] = acc[t][i] - mx;
          }
        }
      sum += __shfl_xor(sum, 32);
      const float ls = logf(sum);
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = SM == 1 ? acc[t][i] / sum : acc[t][i] - ls;
    }
    if (grow < rows) {
      float *yrow = Y + grow * M;
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int f = m0 + 32 * t + 8 * q + 4 * h;
          if ((M & 3) == 0 && f + 3 < M) {
            f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
            *reinterpret_cast<f32x4 *>(yrow + f) = v;
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (f + j < M) yrow[f + j] = acc[t][4 * q + j];
          }
        }
    }
  }
}

template <int MT>
void launch(hipStream_t s, const float *X, const float *W, const float *bias, float *Y, int64_t rows, int K, int M,
            ActParam act, int sm) {
  const int64_t blocks = (rows + 32 * WAVES - 1) / (32 * WAVES);
  const int vec_ok = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  dim3 grid((unsigned)blocks), block(WAVES * 64);
  if (sm == 0) hipLaunchKernelGGL((dense_kernel<MT, 0>), grid, block, 0, s, X, W, bias, Y, rows, K, M, act, vec_ok);
  else if (sm == 1) hipLaunchKernelGGL((dense_kernel<MT, 1>), grid, block, 0, s, X, W, bias, Y, rows, K, M, act, vec_ok);
  else hipLaunchKernelGGL((dense_kernel<MT, 2>), grid, block, 0, s, X, W, bias, Y, rows, K, M, act, vec_ok);
}

}  // namespace

bool dense_can_fuse_softmax(int M) { return M <= 256; }

void dense(hipStream_t s, const float *X, const float *W, const float *bias, float *Y, int64_t rows, int K, int M,
           ActParam act, int softmax_mode) {
  if (rows <= 0) return;
  if (M <= 32) launch<1>(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
  else if (M <= 64) launch<2>(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
  else if (M <= 128) launch<4>(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
  else launch<8>(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
}

}  // namespace infera_hip::kern
