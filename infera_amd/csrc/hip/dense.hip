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

#include <algorithm>
#include <cstdlib>
#include <string_view>

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

  // blockIdx.y selects the group of MT output tiles (a skinny batch with a wide layer, e.g. 256 rows x
  // 1000 classes, would otherwise run on rows/128 workgroups only)
  {
    const int m0 = blockIdx.y * 32 * MT;
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
    {
      float bv[MT][16];  // all bias values requested before the first is used (one memory latency)
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int f = m0 + 32 * t + 8 * (i >> 2) + 4 * h + (i & 3);
          bv[t][i] = (bias != nullptr && f < M) ? bias[f] : 0.f;
        }
      dispatch_act(act.kind, [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
          for (int i = 0; i < 16; i++) acc[t][i] = apply_act_c<KIND>(acc[t][i] + bv[t][i], act.a, act.b);
      });
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

// ---- narrow-output variant: M <= 32, K % 8 == 0 --------------------------------------------------------
// The HBM-bound case (C4: 128 -> 10 + softmax, 552 B/row vs 2,560 flop/row).  No LDS staging of X and no
// block barrier in the row loop: each lane reads its table row in 16-byte pieces directly in B-fragment
// shape (lane half h takes k = 8g+4h..+3, as in mlp_fused.hip), four pieces in flight per wave, many
// waves per CU to cover HBM latency.  The zero-padded 32-column weight fragments (K*128 B) are packed
// into LDS once per block, fragment-major, so one ds_read_b128 feeds four MFMA k-steps.
template <int SM>
__global__ __launch_bounds__(WAVES * 64) void dense_narrow_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                                 const float *__restrict__ bias, float *__restrict__ Y,
                                                                 int64_t rows, int K, int M, ActParam act) {
  extern __shared__ __attribute__((aligned(16))) float wf[];  // [K/8][64 lanes][4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int G = K >> 3;
  for (int i = threadIdx.x; i < G * 256; i += WAVES * 64) {
    const int g = i >> 8, l = (i >> 2) & 63, j = i & 3;
    const int k = 8 * g + 4 * (l >> 5) + j, m = l & 31;
    wf[i] = m < M ? W[int64_t(k) * M + m] : 0.f;
  }
  __syncthreads();
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(wf) + lane;
  float bq[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int f = 8 * (i >> 2) + 4 * h + (i & 3);
    bq[i] = (bias != nullptr && f < M) ? bias[f] : 0.f;
  }
  const int64_t ntiles = (rows + 31) >> 5;
  const int64_t tstride = int64_t(gridDim.x) * WAVES;
  for (int64_t tile = int64_t(blockIdx.x) * WAVES + wave; tile < ntiles; tile += tstride) {
    int64_t row = (tile << 5) + r;
    const bool valid = row < rows;
    if (!valid) row = rows - 1;
    const f32x4 *xp = reinterpret_cast<const f32x4 *>(X + row * K + 4 * h);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    int g = 0;
    for (; g + 4 <= G; g += 4) {
      const f32x4 x0 = xp[2 * g], x1 = xp[2 * g + 2], x2 = xp[2 * g + 4], x3 = xp[2 * g + 6];
      const f32x4 a0 = wq[g * 64], a1 = wq[(g + 1) * 64], a2 = wq[(g + 2) * 64], a3 = wq[(g + 3) * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], x0[j], acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], x1[j], acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], x2[j], acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[j], x3[j], acc, 0, 0, 0);
    }
    for (; g < G; g++) {
      const f32x4 x0 = xp[2 * g];
      const f32x4 a0 = wq[g * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], x0[j], acc, 0, 0, 0);
    }
    // epilogue: lane (r,h) holds features 8*(i>>2) + 4h + (i&3) of table row `row`
    dispatch_act(act.kind, [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
      for (int i = 0; i < 16; i++) acc[i] = apply_act_c<KIND>(acc[i] + bq[i], act.a, act.b);
    });
    if constexpr (SM != 0) {
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (8 * (i >> 2) + 4 * h + (i & 3) < M) mx = fmaxf(mx, acc[i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (8 * (i >> 2) + 4 * h + (i & 3) < M) {
          const float e = expf(acc[i] - mx);
          sum += e;
          acc[i] = SM == 1 ? e : acc[i] - mx;
        }
      sum += __shfl_xor(sum, 32);
      const float ls = logf(sum);
#pragma unroll
      for (int i = 0; i < 16; i++) acc[i] = SM == 1 ? acc[i] / sum : acc[i] - ls;
    }
    if (valid) {
      float *yrow = Y + row * M;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int f = 8 * (i >> 2) + 4 * h + (i & 3);
        if (f < M) yrow[f] = acc[i];
      }
    }
  }
}

// ---- narrowest variant: M <= 16 on v_mfma_f32_16x16x4_f32 ---------------------------------------------------
// Half the matrix-core time of the 32-wide tile for heads like C4's 10 classes (a 32x32 tile would be 69 %
// padding and keeps the MFMA pipe 40 % busy in an HBM-bound kernel).  Lane (n = lane&15, q = lane>>4):
//   B operand: X[row n][16g + 4q + j]   -- one 16-byte load per lane per 16 k, 64 contiguous bytes per row
//   A operand: W[16g + 4q + j][m = lane&15]  (zero-padded, fragment-major in LDS)
//   D: lane holds features 4q..4q+3 of row n.
// A wave owns 32 rows = two 16-row tiles with independent accumulators (the 16x16x4 dependent-issue
// latency of 40 cycles is hidden by alternating them).
using f32x2 = __attribute__((ext_vector_type(2))) float;

// ArgMax epilogue of the 16x16x4 kernels (SM == 3): the row's M scores live in v[0..3] of lanes n, n+16, n+32, n+48
// (feature 4q+i).  Returns the index of the first maximum (ties -> lowest index, like the stand-alone kernel).
__device__ __forceinline__ float argmax_over_quads(const f32x4 &v, int q, int M) {
  float bv = -INFINITY;
  int bi = 4 * q;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (4 * q + i < M && (4 * q + i == 0 || v[i] > bv)) {  // score 0 starts the scan whatever it is (the sequential rule)
      bv = v[i];
      bi = 4 * q + i;
    }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  return float(bi);
}

template <int SM>
__global__ __launch_bounds__(WAVES * 64) void dense_narrow16_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                                   const float *__restrict__ bias, float *__restrict__ Y,
                                                                   int64_t rows, int K, int M, ActParam act) {
  extern __shared__ __attribute__((aligned(16))) float wf[];  // [K/16][64 lanes][4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 15, q = lane >> 4;
  const int G = K >> 4;
  for (int i = threadIdx.x; i < G * 256; i += WAVES * 64) {
    const int g = i >> 8, l = (i >> 2) & 63, j = i & 3;
    const int k = 16 * g + 4 * (l >> 4) + j, m = l & 15;
    wf[i] = m < M ? W[int64_t(k) * M + m] : 0.f;
  }
  __syncthreads();
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(wf) + lane;
  float bq[4];
#pragma unroll
  for (int i = 0; i < 4; i++) bq[i] = (bias != nullptr && 4 * q + i < M) ? bias[4 * q + i] : 0.f;
  const int64_t ntiles = (rows + 31) >> 5;
  const int64_t tstride = int64_t(gridDim.x) * WAVES;
  for (int64_t tile = int64_t(blockIdx.x) * WAVES + wave; tile < ntiles; tile += tstride) {
    int64_t row[2];
    bool valid[2];
    const f32x4 *xp[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      row[t] = (tile << 5) + 16 * t + n;
      valid[t] = row[t] < rows;
      if (!valid[t]) row[t] = rows - 1;
      xp[t] = reinterpret_cast<const f32x4 *>(X + row[t] * K + 4 * q);
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int g = 0;
    // two groups per iteration = 4 row-pieces in flight per wave (four groups measured slower: 5.73 vs
    // 5.40 ms on C4 -- the extra registers cost more occupancy than the extra loads buy)
    for (; g + 2 <= G; g += 2) {
      const f32x4 x00 = xp[0][4 * g], x10 = xp[1][4 * g], x01 = xp[0][4 * g + 4], x11 = xp[1][4 * g + 4];
      const f32x4 a0 = wq[g * 64], a1 = wq[(g + 1) * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x00[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x10[j], acc[1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], x01[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], x11[j], acc[1], 0, 0, 0);
      }
    }
    for (; g < G; g++) {
      const f32x4 x0 = xp[0][4 * g], x1 = xp[1][4 * g];
      const f32x4 a0 = wq[g * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x0[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x1[j], acc[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      f32x4 v = acc[t];
      dispatch_act(act.kind, [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = apply_act_c<KIND>(v[i] + bq[i], act.a, act.b);
      });
      if constexpr (SM == 3) {  // label only: one float per row
        const float label = argmax_over_quads(v, q, M);
        if (valid[t] && q == 0) Y[row[t]] = label;
        continue;
      }
      if constexpr (SM == 1 || SM == 2) {  // the row's features live in lanes n, n+16, n+32, n+48
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (4 * q + i < M) mx = fmaxf(mx, v[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (4 * q + i < M) {
            const float e = expf(v[i] - mx);
            sum += e;
            v[i] = SM == 1 ? e : v[i] - mx;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float ls = logf(sum);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = SM == 1 ? v[i] / sum : v[i] - ls;
      }
      if (valid[t]) {
        float *yrow = Y + row[t] * M + 4 * q;
        if ((M & 1) == 0 && 4 * q + 3 < M) {  // row stride M*4 is 8-byte aligned: two 8-byte stores
          *reinterpret_cast<f32x2 *>(yrow) = f32x2{v[0], v[1]};
          *reinterpret_cast<f32x2 *>(yrow + 2) = f32x2{v[2], v[3]};
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (4 * q + i < M) yrow[i] = v[i];
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
  dim3 grid((unsigned)blocks, unsigned((M + 32 * MT - 1) / (32 * MT))), block(WAVES * 64);
  if constexpr (MT <= 2) {  // softmax epilogues exist for M <= 64 only (the scheduler never asks for more)
    if (sm == 1) { hipLaunchKernelGGL((dense_kernel<MT, 1>), grid, block, 0, s, X, W, bias, Y, rows, K, M, act, vec_ok); return; }
    if (sm == 2) { hipLaunchKernelGGL((dense_kernel<MT, 2>), grid, block, 0, s, X, W, bias, Y, rows, K, M, act, vec_ok); return; }
  }
  hipLaunchKernelGGL((dense_kernel<MT, 0>), grid, block, 0, s, X, W, bias, Y, rows, K, M, act, vec_ok);
}

}  // namespace

// ---- M <= 16, K in {64, 128, 256}: the same 16x16x4 arithmetic fed by COALESCED table reads -----------------
// dense_narrow16_kernel reads X in fragment shape: 64 separate 16-byte accesses per load instruction (the four
// lanes that share a row are 16 lanes apart), each cache line touched by 8 instructions and kept alive in L1
// in between.  Here a wave's 32-row tile is what it is in memory -- one contiguous 128*K-byte run -- and is
// fetched as such (lane l of instruction i takes 16-byte piece 64i + l: the float4-copy access pattern),
// parked in a wave-private LDS region and read back in fragment shape.  Piece (row, c) sits at slot
// row*(K/4) + (c ^ (row & 15)): conflict-free for the linear writes and for the fragment reads (16 lanes =
// 16 rows at one column -> 16 distinct bank quads).  No barrier: the region belongs to one wave, whose LDS
// operations execute in order; the next tile's K/8 loads are in flight (in registers) during the MFMAs.
template <int K, int SM, bool NT>
__global__ __launch_bounds__(WAVES * 64) void dense_narrow16s_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                                    const float *__restrict__ bias, float *__restrict__ Y,
                                                                    int64_t rows, int M, ActParam act, int mode) {
  // mode (wave-uniform): bit 0 = a full tile's 32 x M results are parked in the wave's LDS tile and leave as ONE contiguous run of 16-byte
  // pieces (Y 16-byte aligned).  NT: non-temporal table loads AND result stores (a select between a plain and a non-temporal load of one
  // address folds into the plain one -- it has to be a template parameter)
  constexpr int G = K / 16, PR = K / 4, NL = K / 8;  // k groups, 16-byte pieces per row, load instructions per tile
  static_assert(PR % 16 == 0, "the XOR swizzle needs a multiple of 16 pieces per row");
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [G][64][4] weights, then WAVES x [32 rows][K] tiles
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 15, q = lane >> 4;
  for (int i = threadIdx.x; i < G * 256; i += WAVES * 64) {
    const int g = i >> 8, l = (i >> 2) & 63, j = i & 3;
    const int k = 16 * g + 4 * (l >> 4) + j, m = l & 15;
    smem[i] = m < M ? W[int64_t(k) * M + m] : 0.f;
  }
  __syncthreads();
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(smem) + lane;
  f32x4 *xs = reinterpret_cast<f32x4 *>(smem + G * 256) + wave * (32 * PR);
  float bq[4];
#pragma unroll
  for (int i = 0; i < 4; i++) bq[i] = (bias != nullptr && 4 * q + i < M) ? bias[4 * q + i] : 0.f;
  // where this lane's piece of load instruction i goes, and where its fragments come from
  int wslot[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int p = i * 64 + lane, row = p / PR, c = p % PR;
    wslot[i] = row * PR + (c ^ (row & 15));
  }
  const int64_t ntiles = (rows + 31) >> 5, total4 = rows * PR;
  const int64_t tstride = int64_t(gridDim.x) * WAVES;
  const f32x4 *x4 = reinterpret_cast<const f32x4 *>(X);
  auto fetch = [&](f32x4(&v)[NL], int64_t tile) {
    const int64_t base = tile * (32 * PR) + lane;
    if ((tile + 1) * (32 * PR) <= total4) {  // a whole tile (wave-uniform): NL loads back to back, no per-load bounds branch
#pragma unroll
      for (int i = 0; i < NL; i++) v[i] = NT ? __builtin_nontemporal_load(x4 + base + i * 64) : x4[base + i * 64];
      return;
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int64_t p = base + i * 64;
      v[i] = p < total4 ? x4[p] : f32x4{0.f, 0.f, 0.f, 0.f};  // ragged last tile
    }
  };
  f32x4 stage[NL];
  int64_t tile = int64_t(blockIdx.x) * WAVES + wave;
  if (tile < ntiles) fetch(stage, tile);
  for (; tile < ntiles; tile += tstride) {
#pragma unroll
    for (int i = 0; i < NL; i++) xs[wslot[i]] = stage[i];
    if (tile + tstride < ntiles) fetch(stage, tile + tstride);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int g = 0; g < G; g++) {
      const f32x4 x0 = xs[n * PR + ((4 * g + q) ^ n)], x1 = xs[(16 + n) * PR + ((4 * g + q) ^ n)];
      const f32x4 a0 = wq[g * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x0[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x1[j], acc[1], 0, 0, 0);
      }
    }
    const bool park = SM != 3 && (mode & 1) && (tile << 5) + 32 <= rows;  // (a ragged last tile keeps the per-row stores)
    float *ys = reinterpret_cast<float *>(xs);
    if (park) asm volatile("" ::: "memory");  // the fragment reads above are done with the tile before results are parked over it
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int64_t row = (tile << 5) + 16 * t + n;
      f32x4 v = acc[t];
      dispatch_act(act.kind, [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = apply_act_c<KIND>(v[i] + bq[i], act.a, act.b);
      });
      if constexpr (SM == 3) {  // label only: one float per row
        const float label = argmax_over_quads(v, q, M);
        if (row < rows && q == 0) Y[row] = label;
        continue;
      }
      if constexpr (SM == 1 || SM == 2) {  // the row's features live in lanes n, n+16, n+32, n+48
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (4 * q + i < M) mx = fmaxf(mx, v[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (4 * q + i < M) {
            const float e = expf(v[i] - mx);
            sum += e;
            v[i] = SM == 1 ? e : v[i] - mx;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float ls = logf(sum);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = SM == 1 ? v[i] / sum : v[i] - ls;
      }
      if (park) {
        float *yrow = ys + (16 * t + n) * M + 4 * q;
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (4 * q + i < M) yrow[i] = v[i];
      } else if (row < rows) {
        float *yrow = Y + row * M + 4 * q;
        if ((M & 1) == 0 && 4 * q + 3 < M) {  // row stride M*4 is 8-byte aligned: two 8-byte stores
          if (NT) {
            __builtin_nontemporal_store(f32x2{v[0], v[1]}, reinterpret_cast<f32x2 *>(yrow));
            __builtin_nontemporal_store(f32x2{v[2], v[3]}, reinterpret_cast<f32x2 *>(yrow + 2));
          } else {
            *reinterpret_cast<f32x2 *>(yrow) = f32x2{v[0], v[1]};
            *reinterpret_cast<f32x2 *>(yrow + 2) = f32x2{v[2], v[3]};
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (4 * q + i < M) {
              if (NT) __builtin_nontemporal_store(v[i], yrow + i);
              else yrow[i] = v[i];
            }
        }
      }
    }
    if (park) {  // 32 x M floats = 8M 16-byte pieces, consecutive in Y: lane l takes pieces l, l + 64
      asm volatile("" ::: "memory");
      f32x4 *y4 = reinterpret_cast<f32x4 *>(Y) + tile * (8 * M);
      const f32x4 *ys4 = reinterpret_cast<const f32x4 *>(ys);
      for (int p = lane; p < 8 * M; p += 64) {
        if (NT) __builtin_nontemporal_store(ys4[p], y4 + p);
        else y4[p] = ys4[p];
      }
      asm volatile("" ::: "memory");  // ... before the next tile is parked over the results
    }
  }
}

template <int K>
static void launch_narrow16s(hipStream_t s, const float *X, const float *W, const float *bias, float *Y, int64_t rows, int M,
                             ActParam act, int softmax_mode) {
  const int64_t ntiles = (rows + 31) / 32;
  const size_t lds = (size_t(K) * 16 + size_t(WAVES) * 32 * K) * sizeof(float);
  const int per_cu = int(std::max<size_t>(1, (160 * 1024) / lds));
  int64_t blocks = std::min<int64_t>((ntiles + WAVES - 1) / WAVES, 256 * per_cu);
  dim3 grid((unsigned)blocks), block(WAVES * 64);
  // Round 6 (profiles/r06_c4_store_ab.txt, same process, same buffers, bit-identical): per-row 8-byte stores 5.13 ms per 50M rows; the tile's
  // results parked and written as 16-byte pieces 5.17 (no gain by itself); non-temporal loads + per-row non-temporal stores 5.03; BOTH 4.87 ms =
  // 5.67 TB/s.  INFERA_DENSE16S_MODE=0 / INFERA_DENSE16S_NT=0 (read per launch: measurement + bit-identity test) switch them off.
  const int mode_env = getenv("INFERA_DENSE16S_MODE") ? atoi(getenv("INFERA_DENSE16S_MODE")) : 1;
  const int mode = (reinterpret_cast<uintptr_t>(Y) & 15) == 0 ? mode_env : (mode_env & ~1);
  auto go = [&](auto kernel) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipLaunchKernelGGL(kernel, grid, block, lds, s, X, W, bias, Y, rows, M, act, mode);
  };
  const bool nt = !(getenv("INFERA_DENSE16S_NT") && atoi(getenv("INFERA_DENSE16S_NT")) == 0);
  if (nt) {
    if (softmax_mode == 0) go(dense_narrow16s_kernel<K, 0, true>);
    else if (softmax_mode == 1) go(dense_narrow16s_kernel<K, 1, true>);
    else if (softmax_mode == 2) go(dense_narrow16s_kernel<K, 2, true>);
    else go(dense_narrow16s_kernel<K, 3, true>);
  } else {
    if (softmax_mode == 0) go(dense_narrow16s_kernel<K, 0, false>);
    else if (softmax_mode == 1) go(dense_narrow16s_kernel<K, 1, false>);
    else if (softmax_mode == 2) go(dense_narrow16s_kernel<K, 2, false>);
    else go(dense_narrow16s_kernel<K, 3, false>);
  }
}

// ---- any row length: the 16x16x4 kernel for tables whose rows are not 64/128/256 floats -------------------------
// Real feature tables have 13, 30, 77, 100 columns: rows are not 16-byte aligned and K is no multiple of the MFMA's
// k-group.  The table is still one contiguous run of floats, so a wave fetches its 32 rows (32*K floats, a 128*K-byte
// aligned block) with perfectly coalesced 16-byte loads -- exactly as the bytes lie -- and scatters each float to
// (row, column) of its LDS tile; the tile's rows are KP = 16*ceil(K/16) + 4 floats apart (KP/4 odd: the ds_read_b128
// operand reads of 16 rows hit 16 different bank quads), columns K..16*ceil(K/16) are zeroed once and meet zero
// weights.  From there on it is the narrow16s kernel: B = X[row n][16g+4q+j], A = W fragment-major in LDS, two
// independent 16-row accumulators per wave, next tile's loads in flight during the MFMAs, epilogue in registers.
// NL = 16-byte loads per lane per tile (ceil(K/8)), a compile-time bucket so the prefetch registers are static.
template <int SM, int NL, bool NT>
__global__ __launch_bounds__(WAVES * 64) void dense_narrow16g_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                                    const float *__restrict__ bias, float *__restrict__ Y,
                                                                    int64_t rows, int K, int M, ActParam act, int xflags) {
  // NT: non-temporal table loads + result stores (round 6: 52 -> 16 columns 5.34 -> 5.70 TB/s, other shapes unchanged; parking the results as in
  // dense_narrow16s_kernel LOSES 5-20 % here -- these shapes are bound by their LDS traffic, not by HBM: profiles/r06_dense16g_ab.txt)
  // xflags: bit 0 = X is 16-byte aligned; bit 1 = X is ONE COLUMN-MAJOR chunk [K][rows] (the host path's staging layout: a DataChunk's
  // flat columns as they were copied) -- column k of a tile is 32 consecutive floats at X + k*rows + row0: quad p -> column p/8, rows
  // 4*(p%8)..+3, scattered down a column of the LDS tile; everything after the tile is the row-major kernel.
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [G][64][4] weights, then WAVES x [32 rows][KP] tiles
  const bool x_aligned16 = xflags & 1, xcm = xflags & 2;
  const int G = (K + 15) >> 4, KP = 16 * G + 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 15, q = lane >> 4;
  for (int i = threadIdx.x; i < G * 256; i += WAVES * 64) {
    const int g = i >> 8, l = (i >> 2) & 63, j = i & 3;
    const int k = 16 * g + 4 * (l >> 4) + j, m = l & 15;
    smem[i] = (m < M && k < K) ? W[int64_t(k) * M + m] : 0.f;
  }
  float *xs = smem + G * 256 + wave * (32 * KP);
  for (int i = lane; i < 32 * (KP - K); i += 64) {  // the columns no table float ever lands in
    const int r = i / (KP - K), c = K + i % (KP - K);
    xs[r * KP + c] = 0.f;
  }
  __syncthreads();
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(smem) + lane;
  float bq[4];
#pragma unroll
  for (int i = 0; i < 4; i++) bq[i] = (bias != nullptr && 4 * q + i < M) ? bias[4 * q + i] : 0.f;
  // quad p = i*64 + lane of a tile holds floats 4p..4p+3 of its 32*K: where they go in the LDS tile.  Quads past the
  // tile (the last load instruction is partly idle unless K % 8 == 0) re-read the tile's last quad and park it in the
  // four never-read floats behind row 31, so neither the loads nor the LDS writes need a branch.
  const int nq = 8 * K;
  int slot[NL], col[NL], pq[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int p = i * 64 + lane, e = 4 * p, r = e / K;
    pq[i] = min(p, nq - 1);
    col[i] = p < nq ? e - r * K : 0;
    slot[i] = p < nq ? r * KP + col[i] : 31 * KP + 16 * G;
  }
  const int64_t ntiles = (rows + 31) >> 5, total = rows * K;
  const int64_t full_tiles = x_aligned16 && (!xcm || (rows & 3) == 0) ? rows >> 5 : 0;  // tiles inside the table whose quads are 16-byte aligned
  const int64_t tstride = int64_t(gridDim.x) * WAVES;
  auto fetch = [&](f32x4(&v)[NL], int64_t tile) {
    if (xcm) {
      const int64_t row0 = tile << 5;
      if (tile < full_tiles) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
          const f32x4 *p4 = reinterpret_cast<const f32x4 *>(X + int64_t(pq[i] >> 3) * rows + row0 + 4 * (pq[i] & 7));
          v[i] = NT ? __builtin_nontemporal_load(p4) : *p4;
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < NL; i++) {  // ragged last tile / a row count that is no multiple of 4: element by element
        const int64_t r = row0 + 4 * (pq[i] & 7);
#pragma unroll
        for (int u = 0; u < 4; u++) v[i][u] = r + u < rows ? X[int64_t(pq[i] >> 3) * rows + r + u] : 0.f;
      }
      return;
    }
    if (tile < full_tiles) {  // wave-uniform: straight-line loads, all in flight together
      const f32x4 *src = reinterpret_cast<const f32x4 *>(X + tile * 32 * K);
#pragma unroll
      for (int i = 0; i < NL; i++) v[i] = NT ? __builtin_nontemporal_load(src + pq[i]) : src[pq[i]];
      return;
    }
    const int64_t base = tile * 32 * K;  // ragged end of the table / unaligned base: element by element
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int64_t e = base + 4 * int64_t(pq[i]);
#pragma unroll
      for (int u = 0; u < 4; u++) v[i][u] = e + u < total ? X[e + u] : 0.f;
    }
  };
  f32x4 stage[NL];
  int64_t tile = int64_t(blockIdx.x) * WAVES + wave;
  if (tile < ntiles) fetch(stage, tile);
  const float *x0p = xs + n * KP + 4 * q, *x1p = xs + (16 + n) * KP + 4 * q;
  for (; tile < ntiles; tile += tstride) {
    if (xcm) {
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int p = i * 64 + lane;
#pragma unroll
        for (int u = 0; u < 4; u++) xs[p < nq ? (4 * (p & 7) + u) * KP + (p >> 3) : 31 * KP + 16 * G + u] = stage[i][u];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NL; i++)
#pragma unroll
        for (int u = 0; u < 4; u++) xs[slot[i] + u + (col[i] + u >= K ? KP - K : 0)] = stage[i][u];  // K >= 4: one wrap at most
    }
    if (tile + tstride < ntiles) fetch(stage, tile + tstride);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // operands of group g+1 are read while the eight MFMAs of group g run
    f32x4 x0 = *reinterpret_cast<const f32x4 *>(x0p), x1 = *reinterpret_cast<const f32x4 *>(x1p), a0 = wq[0];
    for (int g = 0; g < G; g++) {
      const int gn = g + 1 < G ? g + 1 : g;
      const f32x4 nx0 = *reinterpret_cast<const f32x4 *>(x0p + 16 * gn), nx1 = *reinterpret_cast<const f32x4 *>(x1p + 16 * gn);
      const f32x4 na0 = wq[gn * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x0[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x1[j], acc[1], 0, 0, 0);
      }
      x0 = nx0;
      x1 = nx1;
      a0 = na0;
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int64_t row = (tile << 5) + 16 * t + n;
      f32x4 v = acc[t];
      dispatch_act(act.kind, [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = apply_act_c<KIND>(v[i] + bq[i], act.a, act.b);
      });
      if constexpr (SM == 3) {  // label only: one float per row
        const float label = argmax_over_quads(v, q, M);
        if (row < rows && q == 0) Y[row] = label;
        continue;
      }
      if constexpr (SM == 1 || SM == 2) {  // the row's features live in lanes n, n+16, n+32, n+48
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (4 * q + i < M) mx = fmaxf(mx, v[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (4 * q + i < M) {
            const float e = expf(v[i] - mx);
            sum += e;
            v[i] = SM == 1 ? e : v[i] - mx;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float ls = logf(sum);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = SM == 1 ? v[i] / sum : v[i] - ls;
      }
      if (row < rows) {
        float *yrow = Y + row * M + 4 * q;
        if ((M & 1) == 0 && 4 * q + 3 < M) {  // row stride M*4 is 8-byte aligned: two 8-byte stores
          if (NT) {
            __builtin_nontemporal_store(f32x2{v[0], v[1]}, reinterpret_cast<f32x2 *>(yrow));
            __builtin_nontemporal_store(f32x2{v[2], v[3]}, reinterpret_cast<f32x2 *>(yrow + 2));
          } else {
            *reinterpret_cast<f32x2 *>(yrow) = f32x2{v[0], v[1]};
            *reinterpret_cast<f32x2 *>(yrow + 2) = f32x2{v[2], v[3]};
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (4 * q + i < M) {
              if (NT) __builtin_nontemporal_store(v[i], yrow + i);
              else yrow[i] = v[i];
            }
        }
      }
    }
  }
}

// Everything up to 128 columns that the 64/128-column kernel does not take, except the shortest rows with one or two
// outputs (measured: the one-lane-per-row VALU kernel below is a few percent ahead for K <= 32, M <= 2).
static bool narrow16g_ok(int K, int M) {
  static const int min_m = getenv("INFERA_DENSE16G_MIN_M") ? atoi(getenv("INFERA_DENSE16G_MIN_M")) : 3;
  return M >= 1 && M <= 16 && K >= 8 && K <= 128 && K != 64 && K != 128 && (K > 32 || M >= min_m);
}

static void launch_narrow16g(hipStream_t s, const float *X, const float *W, const float *bias, float *Y, int64_t rows, int K, int M,
                             ActParam act, int softmax_mode, bool x_colmajor = false) {
  const int G = (K + 15) / 16, KP = 16 * G + 4;
  const int64_t ntiles = (rows + 31) / 32;
  const size_t lds = (size_t(G) * 256 + size_t(WAVES) * 32 * KP) * sizeof(float);
  const int per_cu = int(std::clamp<size_t>((160 * 1024) / lds, 1, 8));
  const int64_t blocks = std::min<int64_t>((ntiles + WAVES - 1) / WAVES, 256 * per_cu);
  const int aligned = ((reinterpret_cast<uintptr_t>(X) & 15) == 0 ? 1 : 0) | (x_colmajor ? 2 : 0);
  dim3 grid((unsigned)blocks), block(WAVES * 64);
  const bool nt = !(getenv("INFERA_DENSE16S_NT") && atoi(getenv("INFERA_DENSE16S_NT")) == 0);  // (non-temporal loads + stores, as launch_narrow16s; per launch: A/B)
  auto go = [&](auto kernel) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipLaunchKernelGGL(kernel, grid, block, lds, s, X, W, bias, Y, rows, K, M, act, aligned);
  };
  auto by_nl = [&](auto smt) {
    constexpr int SMv = decltype(smt)::value;
    if (nt) {
      if (K <= 32) go(dense_narrow16g_kernel<SMv, 4, true>);
      else if (K <= 64) go(dense_narrow16g_kernel<SMv, 8, true>);
      else go(dense_narrow16g_kernel<SMv, 16, true>);
    } else {
      if (K <= 32) go(dense_narrow16g_kernel<SMv, 4, false>);
      else if (K <= 64) go(dense_narrow16g_kernel<SMv, 8, false>);
      else go(dense_narrow16g_kernel<SMv, 16, false>);
    }
  };
  if (softmax_mode == 0) by_nl(std::integral_constant<int, 0>{});
  else if (softmax_mode == 1) by_nl(std::integral_constant<int, 1>{});
  else if (softmax_mode == 2) by_nl(std::integral_constant<int, 2>{});
  else by_nl(std::integral_constant<int, 3>{});
}

// ---- wide tables of any row length (K > 128 columns that the aligned kernels cannot take: 201, 300, 561, 1000) ----
// Rows this long are contiguous runs of >= 516 bytes, so a wave reads them row by row: lane l takes column 64c + l of
// each of its 32 rows -- 256 contiguous bytes per load instruction at whatever alignment the row has -- parks the
// 32 x 64 chunk in LDS (row stride 68 floats: stride-1 writes, conflict-free ds_read_b128 operand reads) and runs the
// four 16-column k-groups of the chunk on the 16x16x4 MFMA while the next chunk's 32 loads are in flight.  Loads are
// clamped into the table and zeroed by select, never branched around.  Weights: fragment-major in LDS, zero-padded to
// whole chunks.  WV waves share them (8 when they are too big for two 4-wave workgroups per CU).
// BIGK (rows longer than ~1500 floats: 2048- / 4096-dimensional embeddings): the weights no longer fit the LDS in one
// piece, so the workgroup keeps a window of WIN chunks (64 KB) and restages it as its waves move along the row
// together -- two barriers per window; every wave of the block runs the same number of tile trips for that.
// MT = 16-wide output tiles (1: M <= 16, 2: M <= 32 -- 20 / 26 / 32-class heads, 4: M <= 64).
template <int SM, int WV, bool BIGK, int MT>
__global__ __launch_bounds__(WV * 64) void dense_narrow16w_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                                 const float *__restrict__ bias, float *__restrict__ Y,
                                                                 int64_t rows, int K, int M, ActParam act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4*NCH][MT][64][4] weights (or a window), then WV x [32][68] chunks
  constexpr int CS = 68;
  constexpr int WIN = 16 / MT;  // chunks per weight window (BIGK)
  const int NCH = (K + 63) >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = lane & 15, q = lane >> 4;
  // fragment-major weights of chunks [c0, c0 + count) into the front of the LDS: [group][mt][lane][j]
  auto stage_weights = [&](int c0, int count) {
    for (int i = threadIdx.x; i < count * 1024 * MT; i += WV * 64) {
      const int j = i & 3, l = (i >> 2) & 63, mt = (i >> 8) % MT, g = 4 * c0 + (i >> 8) / MT;
      const int k = 16 * g + 4 * (l >> 4) + j, m = 16 * mt + (l & 15);
      smem[i] = (m < M && k < K) ? W[int64_t(k) * M + m] : 0.f;
    }
  };
  if constexpr (!BIGK) {
    stage_weights(0, NCH);
    __syncthreads();
  }
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(smem) + lane;
  float *xs = smem + (BIGK ? WIN : NCH) * 1024 * MT + wave * (32 * CS);
  float bq[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int i = 0; i < 4; i++) bq[mt][i] = (bias != nullptr && 16 * mt + 4 * q + i < M) ? bias[16 * mt + 4 * q + i] : 0.f;
  const int64_t ntiles = (rows + 31) >> 5;
  const int64_t tstride = int64_t(gridDim.x) * WV;
  // (tile, chunk) are wave-uniform: one scalar base per fetch, the per-lane offset walks down the rows by K.  (The
  // opaque asm keeps hipcc from hoisting 32 row addresses per call site out of the tile loop -- it did, and spilled.)
  auto fetch = [&](float(&v)[32], int64_t tile, int c) {
    const int col = 64 * c + lane;
    const bool col_ok = col < K;
    int off = col_ok ? col : K - 1;
    asm volatile("" : "+v"(off));
    const float *base = X + (tile << 5) * K;
    if ((tile << 5) + 32 <= rows) {
#pragma unroll
      for (int r = 0; r < 32; r++) {
        const float u = base[off];
        v[r] = col_ok ? u : 0.f;
        off += K;
      }
    } else {  // last, partial tile
      const int left = int(rows - (tile << 5));
#pragma unroll
      for (int r = 0; r < 32; r++) {
        v[r] = (col_ok && r < left) ? base[off] : 0.f;
        off += K;
      }
    }
  };
  float stage[32];
  int64_t tile = int64_t(blockIdx.x) * WV + wave;
  if (tile < ntiles) fetch(stage, tile, 0);
  // BIGK: every wave of the workgroup makes the same number of trips (those past the table only meet the barriers)
  const int64_t first = int64_t(blockIdx.x) * WV;
  for (; BIGK ? first + (tile - first - wave) < ntiles : tile < ntiles; tile += tstride) {
    const bool live = tile < ntiles;
    f32x4 acc[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[0][mt] = acc[1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NCH; c++) {
      if constexpr (BIGK) {
        if (c % WIN == 0) {
          __syncthreads();  // everyone is done with the previous window
          stage_weights(c, min(WIN, NCH - c));
          __syncthreads();
        }
        if (!live) continue;
      }
#pragma unroll
      for (int r = 0; r < 32; r++) xs[r * CS + lane] = stage[r];
      if (c + 1 < NCH) fetch(stage, tile, c + 1);
      else if (tile + tstride < ntiles) fetch(stage, tile + tstride, 0);
#pragma unroll
      for (int gg = 0; gg < 4; gg++) {
        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xs + n * CS + 16 * gg + 4 * q);
        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(xs + (16 + n) * CS + 16 * gg + 4 * q);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const f32x4 a0 = wq[((4 * (BIGK ? c % WIN : c) + gg) * MT + mt) * 64];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x0[j], acc[0][mt], 0, 0, 0);
            acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], x1[j], acc[1][mt], 0, 0, 0);
          }
        }
      }
    }
    if (BIGK && !live) continue;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int64_t row = (tile << 5) + 16 * t + n;
      f32x4 v[MT];
      dispatch_act(act.kind, [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int i = 0; i < 4; i++) v[mt][i] = apply_act_c<KIND>(acc[t][mt][i] + bq[mt][i], act.a, act.b);
      });
      if constexpr (SM == 3) {  // label only: this lane's scores 16mt + 4q + i in ascending order, then the four q groups
        float bv = -INFINITY;
        int bi = M;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int f = 16 * mt + 4 * q + i;
            if (f < M && (f == 0 || v[mt][i] > bv)) {  // score 0 starts the scan whatever it is (the sequential rule)
              bv = v[mt][i];
              bi = f;
            }
          }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          const float ov = __shfl_xor(bv, o);
          const int oi = __shfl_xor(bi, o);
          if (oi < M && (bi == M || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
          }
        }
        if (row < rows && q == 0) Y[row] = float(bi);
        continue;
      }
      if constexpr (SM == 1 || SM == 2) {  // the row's scores live in this lane's MT quads and in lanes n+16, n+32, n+48
        float mx = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (16 * mt + 4 * q + i < M) mx = fmaxf(mx, v[mt][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (16 * mt + 4 * q + i < M) {
              const float e = expf(v[mt][i] - mx);
              sum += e;
              v[mt][i] = SM == 1 ? e : v[mt][i] - mx;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float ls = logf(sum);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int i = 0; i < 4; i++) v[mt][i] = SM == 1 ? v[mt][i] / sum : v[mt][i] - ls;
      }
      if (row < rows) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          float *yrow = Y + row * M + 16 * mt + 4 * q;
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (16 * mt + 4 * q + i < M) yrow[i] = v[mt][i];
        }
      }
    }
  }
}

static size_t narrow16w_lds(int K, int waves, int mt) {
  return (size_t((K + 63) / 64) * 1024 * mt + size_t(waves) * 32 * 68) * sizeof(float);
}
static size_t narrow16w_big_lds(int waves) { return (size_t(16) * 1024 + size_t(waves) * 32 * 68) * sizeof(float); }
static bool narrow16w_ok(int K, int M) { return M >= 1 && M <= 64 && K > 128 && K <= (1 << 20); }

static void launch_narrow16w(hipStream_t s, const float *X, const float *W, const float *bias, float *Y, int64_t rows, int K, int M,
                             ActParam act, int softmax_mode) {
  const int mt = M <= 16 ? 1 : M <= 32 ? 2 : 4;
  const bool big = narrow16w_lds(K, 8, mt) > 160 * 1024;  // weights in 64 KB windows
  const bool eight = big || (2 * narrow16w_lds(K, 4, mt) > 160 * 1024);
  const int waves = eight ? 8 : 4;
  const size_t lds = big ? narrow16w_big_lds(waves) : narrow16w_lds(K, waves, mt);
  const int64_t ntiles = (rows + 31) / 32;
  const int per_cu = int(std::clamp<size_t>((160 * 1024) / lds, 1, 8));
  const int64_t blocks = std::min<int64_t>((ntiles + waves - 1) / waves, 256 * per_cu);
  dim3 grid((unsigned)blocks), block(unsigned(waves) * 64);
  auto go = [&](auto kernel) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipLaunchKernelGGL(kernel, grid, block, lds, s, X, W, bias, Y, rows, K, M, act);
  };
  auto by_w = [&](auto smt, auto mtt) {
    constexpr int SMv = decltype(smt)::value, MTv = decltype(mtt)::value;
    if (big) go(dense_narrow16w_kernel<SMv, 8, true, MTv>);
    else if (eight) go(dense_narrow16w_kernel<SMv, 8, false, MTv>);
    else go(dense_narrow16w_kernel<SMv, 4, false, MTv>);
  };
  auto by_mt = [&](auto smt) {
    if (mt == 1) by_w(smt, std::integral_constant<int, 1>{});
    else if (mt == 2) by_w(smt, std::integral_constant<int, 2>{});
    else by_w(smt, std::integral_constant<int, 4>{});
  };
  if (softmax_mode == 0) by_mt(std::integral_constant<int, 0>{});
  else if (softmax_mode == 1) by_mt(std::integral_constant<int, 1>{});
  else if (softmax_mode == 2) by_mt(std::integral_constant<int, 2>{});
  else by_mt(std::integral_constant<int, 3>{});
}

// ---- skinny layers: rows of at most 32 floats, M <= 16 outputs ----------------------------------------------------
// The most common in-database models are a handful of multiply-adds per row: no matrix core can help, and the
// MFMA kernels' 16-byte operand loads do not even apply (rows of 3, 13, 30 floats are not 16-byte aligned).
// The table is streamed exactly as it lies in memory -- R rows = one contiguous R*K-float run, fetched with
// perfectly coalesced loads into LDS (row stride K|1: conflict-free) -- and each lane then owns one row: a
// k-ordered fmaf chain per output with the weights broadcast from LDS (the oracle's own summation order, so
// results are bit-identical), bias, activation and the optional row softmax / argmax in registers.
template <int MMAX, int SM, int R>
__global__ __launch_bounds__(R) void dense_skinny_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float *__restrict__ Y, int64_t rows, int K,
                                                         int M, ActParam act, int xflags) {
  // xflags: bit 0 = X is 16-byte aligned; bit 1 = X is one column-major chunk [K][rows] (host path): a lane then reads its own row
  // straight from memory -- element k of 256 consecutive rows is one contiguous 1 KB run -- and nothing is staged in LDS.
  const bool x_aligned16 = xflags & 1, xcm = xflags & 2;
  extern __shared__ __attribute__((aligned(16))) float sk[];  // [K][MMAX] weights (zero-padded), [MMAX] bias, [R][KS] rows, 4 spare
  const int KS = K | 1;
  float *wl = sk, *bl = sk + K * MMAX, *xs = bl + MMAX;
  for (int i = threadIdx.x; i < K * MMAX; i += R) {
    const int k = i / MMAX, m = i - k * MMAX;
    wl[i] = m < M ? W[k * M + m] : 0.f;
  }
  if (threadIdx.x < MMAX) bl[threadIdx.x] = (bias != nullptr && int(threadIdx.x) < M) ? bias[threadIdx.x] : 0.f;
  const int64_t ntiles = (rows + R - 1) / R, total = rows * K;
  const int64_t full_tiles = x_aligned16 ? rows / R : 0;
  // A tile is R*K floats = R*K/4 quads (16-byte aligned: R*K*4 bytes per tile, R a multiple of 4); thread t fetches
  // quads t, t+R, ... (K <= 32: at most eight, all in flight together), then scatters the four floats of each to
  // (row, column).  Threads past the tile's last quad re-read it and park it in the spare floats: no branches.
  constexpr int NQ = 8;
  const int nq = (R / 4) * K, nj = (K + 3) / 4;  // nj = load instructions a tile needs (wave-uniform)
  int pq[NQ], r0[NQ], c0[NQ];
#pragma unroll
  for (int j = 0; j < NQ; j++) {
    const int q = int(threadIdx.x) + j * R;
    pq[j] = min(q, nq - 1);
    r0[j] = (4 * q) / K;
    c0[j] = 4 * q - r0[j] * K;
    if (q >= nq) r0[j] = c0[j] = -8;  // parked: stays negative through the four column steps below
  }
  if (xcm) __syncthreads();  // weights visible (the row-major path has its own barriers per tile)
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * R * K;
    f32x4 v[NQ];
    if (xcm) {
    } else if (tile < full_tiles) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(X + base);
#pragma unroll
      for (int j = 0; j < NQ; j++)
        if (j < nj) v[j] = src[pq[j]];
    } else {  // ragged end of the table / unaligned base
#pragma unroll
      for (int j = 0; j < NQ; j++) {
        if (j >= nj) break;
        const int64_t e = base + 4 * int64_t(pq[j]);
#pragma unroll
        for (int u = 0; u < 4; u++) v[j][u] = e + u < total ? X[e + u] : 0.f;
      }
    }
    if (!xcm) {
      __syncthreads();  // weights visible (first trip) / previous tile's rows consumed
#pragma unroll
      for (int j = 0; j < NQ; j++) {
        if (j >= nj) break;
        int r = r0[j], c = c0[j];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          xs[c < 0 ? R * KS + u : r * KS + c] = v[j][u];
          if (++c == K) {
            c = 0;
            r++;
          }
        }
      }
      __syncthreads();
    }
    float acc[MMAX];
#pragma unroll
    for (int m = 0; m < MMAX; m++) acc[m] = 0.f;
    const int lrow = int(threadIdx.x);
    if (xcm) {  // the same k-ordered fmaf chain, the row's elements eight at a time straight from the column-major chunk
      const int64_t grow = min(tile * R + lrow, rows - 1);  // (lanes past the table re-read its last row; nothing is stored for them)
      const float *xc = X + grow;
      auto chain = [&](auto n_tag) {  // ALL of the row's elements requested before the first is used: one memory round trip per tile
        constexpr int NK = decltype(n_tag)::value;
        float xv[NK];
#pragma unroll
        for (int u = 0; u < NK; u++) xv[u] = xc[int64_t(min(u, K - 1)) * rows];
#pragma unroll
        for (int u = 0; u < NK; u++) {
          if (u >= K) break;
#pragma unroll
          for (int m = 0; m < MMAX; m++) acc[m] = fmaf(xv[u], wl[u * MMAX + m], acc[m]);
        }
      };
      if (K <= 8) chain(std::integral_constant<int, 8>{});
      else if (K <= 16) chain(std::integral_constant<int, 16>{});
      else chain(std::integral_constant<int, 32>{});
    } else {
      const float *xr = xs + lrow * KS;
      for (int k = 0; k < K; k++) {
        const float x = xr[k];
#pragma unroll
        for (int m = 0; m < MMAX; m++) acc[m] = fmaf(x, wl[k * MMAX + m], acc[m]);
      }
    }
    dispatch_act(act.kind, [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
      for (int m = 0; m < MMAX; m++) acc[m] = apply_act_c<KIND>(acc[m] + bl[m], act.a, act.b);
    });
    const int64_t row = tile * R + lrow;
    if constexpr (SM == 3) {  // ArgMax over the M scores: the label is the only thing written
      float best = acc[0];
      int bi = 0;
#pragma unroll
      for (int m = 1; m < MMAX; m++)
        if (m < M && acc[m] > best) {
          best = acc[m];
          bi = m;
        }
      if (row < rows) Y[row] = float(bi);
      continue;
    }
    if constexpr (SM == 1 || SM == 2) {
      float mx = -INFINITY;
#pragma unroll
      for (int m = 0; m < MMAX; m++)
        if (m < M) mx = fmaxf(mx, acc[m]);
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < MMAX; m++)
        if (m < M) {
          const float e = expf(acc[m] - mx);
          sum += e;
          acc[m] = SM == 1 ? e : acc[m] - mx;
        }
      const float ls = logf(sum);
#pragma unroll
      for (int m = 0; m < MMAX; m++) acc[m] = SM == 1 ? acc[m] / sum : acc[m] - ls;
    }
    if (row < rows) {
      float *y = Y + row * M;
      if (MMAX >= 4 && M == MMAX && (reinterpret_cast<uintptr_t>(Y) & 15) == 0) {  // whole quads, 16-byte aligned rows
#pragma unroll
        for (int m = 0; m < MMAX; m += 4) *reinterpret_cast<f32x4 *>(y + m) = f32x4{acc[m], acc[m + 1], acc[m + 2], acc[m + 3]};
      } else {
#pragma unroll
        for (int m = 0; m < MMAX; m++)
          if (m < M) y[m] = acc[m];
      }
    }
  }
}

// rows of up to 32 floats (wider ones go to the 16x16x4 kernels above)
static bool skinny_ok(int K, int M) { return M >= 1 && M <= 16 && K >= 1 && K <= 32; }

static void launch_skinny(hipStream_t s, const float *X, const float *W, const float *bias, float *Y, int64_t rows, int K, int M,
                          ActParam act, int softmax_mode, bool x_colmajor = false) {
  const int mmax = M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : M <= 8 ? 8 : 16;
  constexpr int R = 256;  // rows per workgroup, one thread each
  const size_t lds = (size_t(K) * mmax + mmax + size_t(R) * size_t(K | 1) + 4) * sizeof(float);
  const int64_t ntiles = (rows + R - 1) / R;
  const unsigned grid = unsigned(std::min<int64_t>(ntiles, 256 * 8));
  const int aligned = ((reinterpret_cast<uintptr_t>(X) & 15) == 0 ? 1 : 0) | (x_colmajor ? 2 : 0);
  auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(R), lds, s, X, W, bias, Y, rows, K, M, act, aligned); };
  auto by_sm = [&](auto mm) {
    constexpr int MM = decltype(mm)::value;
    if (softmax_mode == 0) go(dense_skinny_kernel<MM, 0, R>);
    else if (softmax_mode == 1) go(dense_skinny_kernel<MM, 1, R>);
    else if (softmax_mode == 2) go(dense_skinny_kernel<MM, 2, R>);
    else go(dense_skinny_kernel<MM, 3, R>);
  };
  switch (mmax) {
    case 1: by_sm(std::integral_constant<int, 1>{}); break;
    case 2: by_sm(std::integral_constant<int, 2>{}); break;
    case 4: by_sm(std::integral_constant<int, 4>{}); break;
    case 8: by_sm(std::integral_constant<int, 8>{}); break;
    default: by_sm(std::integral_constant<int, 16>{}); break;
  }
}

bool dense_can_fuse_softmax(int K, int M) { return M <= 16 || (M <= 32 && K % 8 == 0 && K <= 512) || (M <= 64 && K > 128); }
// ArgMax epilogues (softmax_mode 3) exist in the skinny and the two 16x16x4 streaming kernels; the latter need 16-byte rows
bool dense_can_fuse_argmax(const float *X, int K, int M) {
  return narrow16g_ok(K, M) || skinny_ok(K, M) || narrow16w_ok(K, M) ||
         (M <= 16 && K % 16 == 0 && K <= 1024 && (reinterpret_cast<uintptr_t>(X) & 15) == 0);
}

// Layers whose kernels can read a column-major chunk (the host path's staging layout) themselves: the two as-it-lies streaming
// kernels, up to 63 columns and 16 outputs -- the linear / logistic regressions and small classifiers of the tabular world.
// (64 columns and more: the chunk is 0.5 MB+, one more tiny launch no longer shows, and the transposed chunk feeds the faster
// aligned kernels -- measured equal or a few percent behind on 128 -> 10)
bool dense_colmajor_supported(int K, int M) { return M >= 1 && M <= 16 && K >= 1 && K < 64; }

// Which kernel family serves a Dense layer: ONE decision function, used by the launcher below and by the plan description
// (infera_hip_get_plan "dense_kernels": what profiles/traffic_*.json and the rocprofv3 summaries must name).
const char *dense_kernel_family(int64_t rows, int K, int M, int softmax_mode, bool x_colmajor, bool aligned16) {
  if (x_colmajor) return K >= 8 && (K > 32 || M >= 3) ? "dense_narrow16g_kernel" : "dense_skinny_kernel";
  static const int wide16 = getenv("INFERA_DENSE16W") ? atoi(getenv("INFERA_DENSE16W")) : 1;  // 0 off, 2 = also where aligned kernels exist (A/B)
  if (wide16 == 2 && M <= 16 && K >= 64) return "dense_narrow16w_kernel";
  if (narrow16g_ok(K, M)) return "dense_narrow16g_kernel";
  if (skinny_ok(K, M)) return "dense_skinny_kernel";
  static const bool staged16 = !(getenv("INFERA_DENSE16_STAGED") && atoi(getenv("INFERA_DENSE16_STAGED")) == 0);
  if (staged16 && M <= 16 && (K == 64 || K == 128 || K == 256) && rows >= 4096 && aligned16) return "dense_narrow16s_kernel";
  if (M <= 16 && K % 16 == 0 && K <= 1024 && aligned16) return "dense_narrow16_kernel";
  if (wide16 && narrow16w_ok(K, M)) return "dense_narrow16w_kernel";
  if (softmax_mode == 3) return "";  // callers check dense_can_fuse_argmax first; nothing below has that epilogue
  if (M <= 32 && K % 8 == 0 && K <= 512 && aligned16) return "dense_narrow_kernel";
  return "dense_kernel";
}

void dense(hipStream_t s, const float *X, const float *W, const float *bias, float *Y, int64_t rows, int K, int M,
           ActParam act, int softmax_mode, bool x_colmajor) {
  if (rows <= 0) return;
  const std::string_view fam = dense_kernel_family(rows, K, M, softmax_mode, x_colmajor, (reinterpret_cast<uintptr_t>(X) & 15) == 0);
  if (fam.empty()) return;
  if (fam == "dense_narrow16g_kernel") return launch_narrow16g(s, X, W, bias, Y, rows, K, M, act, softmax_mode, x_colmajor);
  if (fam == "dense_skinny_kernel") return launch_skinny(s, X, W, bias, Y, rows, K, M, act, softmax_mode, x_colmajor);
  if (fam == "dense_narrow16w_kernel") return launch_narrow16w(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
  if (fam == "dense_narrow16s_kernel") {
    if (K == 64) launch_narrow16s<64>(s, X, W, bias, Y, rows, M, act, softmax_mode);
    else if (K == 128) launch_narrow16s<128>(s, X, W, bias, Y, rows, M, act, softmax_mode);
    else launch_narrow16s<256>(s, X, W, bias, Y, rows, M, act, softmax_mode);
    return;
  }
  if (fam == "dense_narrow16_kernel") {
    const int64_t ntiles = (rows + 31) / 32;
    int64_t blocks = (ntiles + WAVES - 1) / WAVES;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const size_t lds = size_t(K) * 16 * sizeof(float);
    dim3 grid((unsigned)blocks), block(WAVES * 64);
    if (softmax_mode == 0) hipLaunchKernelGGL((dense_narrow16_kernel<0>), grid, block, lds, s, X, W, bias, Y, rows, K, M, act);
    else if (softmax_mode == 1) hipLaunchKernelGGL((dense_narrow16_kernel<1>), grid, block, lds, s, X, W, bias, Y, rows, K, M, act);
    else if (softmax_mode == 2) hipLaunchKernelGGL((dense_narrow16_kernel<2>), grid, block, lds, s, X, W, bias, Y, rows, K, M, act);
    else hipLaunchKernelGGL((dense_narrow16_kernel<3>), grid, block, lds, s, X, W, bias, Y, rows, K, M, act);
    return;
  }
  if (fam == "dense_narrow_kernel") {
    const int64_t ntiles = (rows + 31) / 32;
    int64_t blocks = (ntiles + WAVES - 1) / WAVES;
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride beyond 8 blocks per CU
    const size_t lds = size_t(K) * 32 * sizeof(float);
    dim3 grid((unsigned)blocks), block(WAVES * 64);
    if (softmax_mode == 0) hipLaunchKernelGGL((dense_narrow_kernel<0>), grid, block, lds, s, X, W, bias, Y, rows, K, M, act);
    else if (softmax_mode == 1) hipLaunchKernelGGL((dense_narrow_kernel<1>), grid, block, lds, s, X, W, bias, Y, rows, K, M, act);
    else hipLaunchKernelGGL((dense_narrow_kernel<2>), grid, block, lds, s, X, W, bias, Y, rows, K, M, act);
    return;
  }
  // MT = output tiles per workgroup: as many as the layer has (X is staged once per workgroup), but never
  // so many that a small batch leaves most of the 256 CUs idle.
  const int64_t row_blocks = (rows + 32 * WAVES - 1) / (32 * WAVES);
  int mt = M <= 32 ? 1 : M <= 64 ? 2 : M <= 128 ? 4 : 8;
  if (softmax_mode == 0)
    while (mt > 1 && row_blocks * ((M + 32 * mt - 1) / (32 * mt)) < 512) mt >>= 1;
  if (mt == 1) launch<1>(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
  else if (mt == 2) launch<2>(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
  else if (mt == 4) launch<4>(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
  else launch<8>(s, X, W, bias, Y, rows, K, M, act, softmax_mode);
}

}  // namespace infera_hip::kern
