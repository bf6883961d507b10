// E2: superpoint matching scores (SuperPointMatching.forward, geotransformer/modules/geotransformer/superpoint_matching.py:31-39):
//   S[n, m] = exp(-clamp(2 - 2 f_ref[n].f_src[m], 0))  on L2-normalised features, then the dual normalisation
//   S / rowsum(S) * S / colsum(S).  Three small launches (scores + row sums, column sums, normalise); deterministic.
#include "common.h"

namespace {

// one workgroup per reference row n; thread t handles columns t, t + 256, ...
__global__ __launch_bounds__(256) void sp_scores_kernel(const float* __restrict__ ref, const float* __restrict__ src, int N,
                                                        int M, int C, float* __restrict__ S, float* __restrict__ rowsum) {
  extern __shared__ float rf[];          // C floats of the reference row + 4 partials
  __shared__ float part[4];
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) rf[c] = ref[(size_t)n * C + c];
  __syncthreads();
  float rs = 0.f;
  for (int m = threadIdx.x; m < M; m += 256) {
    const float4* sp = reinterpret_cast<const float4*>(src + (size_t)m * C);
    float acc = 0.f;
    for (int c4 = 0; c4 < C / 4; c4++) {
      const float4 v = sp[c4];
      acc += (rf[4 * c4] * v.x + rf[4 * c4 + 1] * v.y) + (rf[4 * c4 + 2] * v.z + rf[4 * c4 + 3] * v.w);
    }
    const float s = __expf(-fmaxf(2.f - 2.f * acc, 0.f));
    S[(size_t)n * M + m] = s;
    rs += s;
  }
  rs = se3_wave_sum(rs);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = rs;
  __syncthreads();
  if (threadIdx.x == 0) rowsum[n] = (part[0] + part[1]) + (part[2] + part[3]);
}

// column sums: 64 columns x 4 row lanes per workgroup (the single-thread-per-column version was one serial pass over N rows on
// ceil(M / 256) workgroups: 89 us at N = M = 382); fixed summation order -> deterministic
__global__ __launch_bounds__(256) void sp_colsum_kernel(const float* __restrict__ S, int N, int M, float* __restrict__ colsum) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + cl;
  float cs = 0.f;
  if (m < M)
    for (int n = rl; n < N; n += 4) cs += S[(size_t)n * M + m];
  part[rl][cl] = cs;
  __syncthreads();
  if (rl == 0 && m < M) colsum[m] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

__global__ void sp_normalize_kernel(float* __restrict__ S, const float* __restrict__ rowsum,
                                    const float* __restrict__ colsum, int N, int M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * M) return;
  const int n = (int)(i / M), m = (int)(i - (int64_t)n * M);
  const float s = S[i];
  S[i] = (s / rowsum[n]) * (s / colsum[m]);
}

}  // namespace

extern "C" int se3_superpoint_scores(const float* ref_feats, const float* src_feats, int N, int M, int C,
                                     int dual_normalization, float* scores, float* workspace, void* stream) {
  SE3_REQUIRE(ref_feats && src_feats && scores && workspace, SE3_ERR_INVALID_ARG, "superpoint_scores: null pointer");
  SE3_REQUIRE(N >= 1 && M >= 1 && C >= 4 && C % 4 == 0 && C <= 4096, SE3_ERR_UNSUPPORTED, "superpoint_scores: N %d M %d C %d", N, M, C);
  hipStream_t st = (hipStream_t)stream;
  float* rowsum = workspace;
  float* colsum = workspace + N;
  sp_scores_kernel<<<N, 256, C * sizeof(float), st>>>(ref_feats, src_feats, N, M, C, scores, rowsum);
  if (dual_normalization) {
    sp_colsum_kernel<<<(M + 63) / 64, 256, 0, st>>>(scores, N, M, colsum);
    const int64_t total = (int64_t)N * M;
    sp_normalize_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(scores, rowsum, colsum, N, M);
  }
  SE3_CHECK_LAUNCH("superpoint_scores");
  return SE3_OK;
}
