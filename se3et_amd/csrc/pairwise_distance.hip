// E1: squared pairwise distances of two row sets (geotransformer/modules/ops/pairwise_distance.py:4-30):
//   out[b, n, m] = max(|x[b,n]|^2 - 2 x[b,n].y[b,m] + |y[b,m]|^2, 0)      (normalized: max(2 - 2 x.y, 0), unit vectors)
// One workgroup = a 64 x 64 block of the output, a wave = 32 x 32 of it as 2 x 2 tiles of v_mfma_f32_16x16x4_f32 (f32 operands: the result
// feeds thresholds and square roots, and the products are the small part of the sum when the points are close).  A lane's float4 holds four
// consecutive channels k0 + 4 kq .. + 3 of its row; MFMA i of a 16-channel step takes component i of both operands (the same permutation of
// the K index on both sides), so a step is 4 loads and 16 MFMAs per wave.  The squared norms come from the same registers.  Any channel count
// (3-D points: C = 3) through the scalar-load instantiation; rows are clamped, not branched on.
#include "common.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool VEC>
__device__ __forceinline__ f32x4 pd_load4(const float* row, int k, int C) {
  if (VEC) return k < C ? *reinterpret_cast<const f32x4*>(row + k) : f32x4{0.f, 0.f, 0.f, 0.f};      // C % 4 == 0: k < C covers k + 3
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = k + i < C ? row[k + i] : 0.f;
  return v;
}

template <bool VEC>
__global__ __launch_bounds__(256) void pairwise_distance_kernel(const float* __restrict__ x, const float* __restrict__ y, int N, int M, int C,
                                                                int64_t x_bs, int64_t y_bs, int normalized, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  const int64_t b = blockIdx.z;
  const int row0 = blockIdx.y * 64 + (wave >> 1) * 32, col0 = blockIdx.x * 64 + (wave & 1) * 32;
  if (row0 >= N || col0 >= M) return;
  const float* xr[2];
  const float* yr[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    xr[t] = x + b * x_bs + (int64_t)min(row0 + 16 * t + c, N - 1) * C;
    yr[t] = y + b * y_bs + (int64_t)min(col0 + 16 * t + c, M - 1) * C;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float x2[2] = {0.f, 0.f}, y2[2] = {0.f, 0.f};
  f32x4 a[2], bq[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    a[t] = pd_load4<VEC>(xr[t], 4 * kq, C);
    bq[t] = pd_load4<VEC>(yr[t], 4 * kq, C);
  }
  for (int k0 = 0; k0 < C; k0 += 16) {
    f32x4 an[2], bn[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {                              // the next step's rows behind this step's MFMAs (past the end: zeros)
      an[t] = pd_load4<VEC>(xr[t], k0 + 16 + 4 * kq, C);
      bn[t] = pd_load4<VEC>(yr[t], k0 + 16 + 4 * kq, C);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      x2[t] += (a[t][0] * a[t][0] + a[t][1] * a[t][1]) + (a[t][2] * a[t][2] + a[t][3] * a[t][3]);
      y2[t] += (bq[t][0] * bq[t][0] + bq[t][1] * bq[t][1]) + (bq[t][2] * bq[t][2] + bq[t][3] * bq[t][3]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti][i], bq[tj][i], acc[ti][tj], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) {
      a[t] = an[t];
      bq[t] = bn[t];
    }
  }
#pragma unroll
  for (int t = 0; t < 2; t++) {                                // the four channel quarters of a row live in lanes c, c + 16, c + 32, c + 48
    x2[t] += __shfl_xor(x2[t], 16);
    x2[t] += __shfl_xor(x2[t], 32);
    y2[t] += __shfl_xor(y2[t], 16);
    y2[t] += __shfl_xor(y2[t], 32);
  }
  // acc[ti][tj][r] = x[row0 + 16 ti + 4 kq + r] . y[col0 + 16 tj + c]
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = row0 + 16 * ti + 4 * kq + r;
      const float xn = __shfl(x2[ti], 4 * kq + r);             // lane 4 kq + r holds the norm of that row
#pragma unroll
      for (int tj = 0; tj < 2; tj++) {
        const int col = col0 + 16 * tj + c;
        const float xy = acc[ti][tj][r];
        float d = normalized ? 2.0f - 2.0f * xy : (xn - 2.0f * xy) + y2[tj];
        d = d < 0.f ? 0.f : d;                                 // clamp(min = 0); a NaN stays a NaN, as in the reference
        if (row < N && col < M) out[(b * N + row) * M + col] = d;
      }
    }
}
}  // namespace

extern "C" int se3_pairwise_distance(const float* x, const float* y, int64_t batch, int N, int M, int C, int64_t x_batch_stride,
                                     int64_t y_batch_stride, int normalized, float* out, void* stream) {
  SE3_REQUIRE(batch >= 0 && N >= 0 && M >= 0 && C >= 1, SE3_ERR_INVALID_ARG, "pairwise_distance: batch %lld N %d M %d C %d", (long long)batch, N, M, C);
  if (batch == 0 || N == 0 || M == 0) return SE3_OK;
  SE3_REQUIRE(x && y && out, SE3_ERR_INVALID_ARG, "pairwise_distance: null pointer");
  SE3_REQUIRE(batch <= 65535, SE3_ERR_UNSUPPORTED, "pairwise_distance: %lld batches (at most 65535)", (long long)batch);
  const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64), (unsigned)batch);
  const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && x_batch_stride % 4 == 0 &&
                   y_batch_stride % 4 == 0;
  hipStream_t st = (hipStream_t)stream;
  if (vec) pairwise_distance_kernel<true><<<grid, 256, 0, st>>>(x, y, N, M, C, x_batch_stride, y_batch_stride, normalized, out);
  else pairwise_distance_kernel<false><<<grid, 256, 0, st>>>(x, y, N, M, C, x_batch_stride, y_batch_stride, normalized, out);
  SE3_CHECK_LAUNCH("pairwise_distance");
  return SE3_OK;
}
