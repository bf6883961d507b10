// E2: superpoint matching scores (SuperPointMatching.forward, geotransformer/modules/geotransformer/superpoint_matching.py:31-39):
//   S[n, m] = exp(-clamp(2 - 2 f_ref[n].f_src[m], 0))  on L2-normalised features, then the dual normalisation
//   S / rowsum(S) * S / colsum(S).  Three small launches (scores + row sums, column sums, normalise); deterministic.
#include "common.h"

namespace {

// one K4-step of a feature dot product and the score of a finished one, with the roundings spelled out: the per-pair kernel and the stack
// kernel (eight rows per workgroup) must agree bit for bit, whatever the compiler would contract in either loop
__device__ __forceinline__ float sp_dot_step(float acc, const float* f, const float4& v) {
  const float a = __fmaf_rn(f[0], v.x, __fmul_rn(f[1], v.y));
  const float b = __fmaf_rn(f[2], v.z, __fmul_rn(f[3], v.w));
  return __fadd_rn(acc, __fadd_rn(a, b));
}
__device__ __forceinline__ float sp_score(float dot) { return __expf(-fmaxf(__fmaf_rn(-2.f, dot, 2.f), 0.f)); }

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
      acc = sp_dot_step(acc, rf + 4 * c4, v);
    }
    const float s = sp_score(acc);
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

// ---- stack mode: the superpoint pairs of several registration pairs in one launch per kernel -----------------------------
constexpr int kMaxMatchPairs = 16;
struct MatchPairs {
  int ref_row[kMaxMatchPairs], src_row[kMaxMatchPairs];       // first row of the pair's ref / src superpoints in `feats`
  int N[kMaxMatchPairs], M[kMaxMatchPairs];
  int ref_mask[kMaxMatchPairs], src_mask[kMaxMatchPairs];     // first entry of the pair's node masks in `node_masks`
  int row0[kMaxMatchPairs + 1];                               // prefix sums of N (flat ref-row index -> pair)
  int col0[kMaxMatchPairs + 1];                               // prefix sums of M
  int n;
};

// one workgroup per (pair, 8 reference rows); raw scores into the pair's rows of the padded (B, stride) output, row sums over the
// valid columns into the workspace.  Nodes with mask 0 (no fine point) are absent, as the reference drops them beforehand.
// (Late round 5: eight rows per workgroup instead of one -- a thread's source row (1 KB, one lane per row: 64 cache lines per load
// instruction) is fetched once for eight dot products instead of once per dot product; every dot product keeps its order of operations.)
template <int kSpRows>              // 8; 1 for feature widths whose eight rows would not fit 64 KB of LDS
__global__ __launch_bounds__(256) void sp_scores_stack_kernel(const float* __restrict__ feats, const unsigned char* __restrict__ masks,
                                                              MatchPairs T, int C, int64_t stride, float* __restrict__ S,
                                                              float* __restrict__ rowsum) {
  extern __shared__ float rf[];                                 // kSpRows x C
  __shared__ float part[kSpRows][4];
  const int p = blockIdx.y, N = T.N[p], M = T.M[p];
  const int n0 = blockIdx.x * kSpRows;
  if (n0 >= N) return;
  const int nr = min(kSpRows, N - n0);
  for (int e = threadIdx.x; e < kSpRows * C; e += 256) {
    const int r = e / C, c = e - r * C;
    rf[e] = r < nr ? feats[(size_t)(T.ref_row[p] + n0 + r) * C + c] : 0.f;
  }
  __syncthreads();
  const unsigned char* cm = masks + T.src_mask[p];
  const unsigned char* rmk = masks + T.ref_mask[p] + n0;
  bool present[kSpRows];
#pragma unroll
  for (int r = 0; r < kSpRows; r++) present[r] = r < nr && rmk[r < nr ? r : 0] != 0;
  float rs[kSpRows];
#pragma unroll
  for (int r = 0; r < kSpRows; r++) rs[r] = 0.f;
  for (int m = threadIdx.x; m < M; m += 256) {
    const float4* sp = reinterpret_cast<const float4*>(feats + (size_t)(T.src_row[p] + m) * C);
    float acc[kSpRows];
#pragma unroll
    for (int r = 0; r < kSpRows; r++) acc[r] = 0.f;
    for (int c4 = 0; c4 < C / 4; c4++) {
      const float4 v = sp[c4];
#pragma unroll
      for (int r = 0; r < kSpRows; r++) {
        acc[r] = sp_dot_step(acc[r], rf + r * C + 4 * c4, v);
      }
    }
    const bool cv = cm[m] != 0;
#pragma unroll
    for (int r = 0; r < kSpRows; r++) {
      if (!present[r]) continue;
      const float sc = cv ? sp_score(acc[r]) : 0.f;
      S[(size_t)p * stride + (size_t)(n0 + r) * M + m] = sc;
      rs[r] += sc;
    }
  }
#pragma unroll
  for (int r = 0; r < kSpRows; r++) {
    const float w = se3_wave_sum(rs[r]);
    if ((threadIdx.x & 63) == 0) part[r][threadIdx.x >> 6] = w;
  }
  __syncthreads();
  if ((int)threadIdx.x < nr && rmk[threadIdx.x])
    rowsum[T.row0[p] + n0 + threadIdx.x] = (part[threadIdx.x][0] + part[threadIdx.x][1]) + (part[threadIdx.x][2] + part[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void sp_colsum_stack_kernel(const float* __restrict__ S, const unsigned char* __restrict__ masks,
                                                              MatchPairs T, int64_t stride, float* __restrict__ colsum) {
  __shared__ float part[4][64];
  const int p = blockIdx.y, N = T.N[p], M = T.M[p];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + cl;
  if (blockIdx.x * 64 >= M) return;
  const float* Sp = S + (size_t)p * stride;
  const unsigned char* rm = masks + T.ref_mask[p];
  float cs = 0.f;
  if (m < M)
    for (int n = rl; n < N; n += 4) cs += rm[n] ? Sp[(size_t)n * M + m] : 0.f;
  part[rl][cl] = cs;
  __syncthreads();
  if (rl == 0 && m < M) colsum[T.col0[p] + m] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// dual normalisation in place; absent nodes and the padding beyond N * M get -1 (below every score), so that one top-k over
// the padded rows selects per pair
__global__ void sp_normalize_stack_kernel(float* __restrict__ S, const unsigned char* __restrict__ masks, MatchPairs T,
                                          int64_t stride, const float* __restrict__ rowsum, const float* __restrict__ colsum,
                                          int dual) {
  const int p = blockIdx.y, N = T.N[p], M = T.M[p];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= stride) return;
  float* Sp = S + (size_t)p * stride;
  if (i >= (int64_t)N * M) {
    Sp[i] = -1.f;
    return;
  }
  const int n = (int)(i / M), m = (int)(i - (int64_t)n * M);
  if (!masks[T.ref_mask[p] + n] || !masks[T.src_mask[p] + m]) {
    Sp[i] = -1.f;
    return;
  }
  if (dual) {
    const float s = Sp[i];
    Sp[i] = (s / rowsum[T.row0[p] + n]) * (s / colsum[T.col0[p] + m]);
  }
}

}  // namespace

extern "C" int se3_superpoint_scores_stack(const float* feats, const uint8_t* node_masks, const int64_t* ref_rows,
                                           const int64_t* src_rows, const int64_t* ref_lengths, const int64_t* src_lengths,
                                           const int64_t* ref_mask_offsets, const int64_t* src_mask_offsets, int num_pairs, int C,
                                           int dual_normalization, int64_t score_stride, float* scores, float* workspace,
                                           void* stream) {
  SE3_REQUIRE(feats && node_masks && ref_rows && src_rows && ref_lengths && src_lengths && ref_mask_offsets && src_mask_offsets &&
                  scores && workspace, SE3_ERR_INVALID_ARG, "superpoint_scores_stack: null pointer");
  SE3_REQUIRE(num_pairs >= 1 && num_pairs <= kMaxMatchPairs && C >= 4 && C % 4 == 0 && C <= 4096, SE3_ERR_UNSUPPORTED,
              "superpoint_scores_stack: %d pairs (1..%d), C %d", num_pairs, kMaxMatchPairs, C);
  MatchPairs T{};
  T.n = num_pairs;
  int64_t rows = 0, cols = 0;
  int max_m = 0;
  for (int p = 0; p < num_pairs; p++) {
    SE3_REQUIRE(ref_lengths[p] >= 1 && src_lengths[p] >= 1 && ref_lengths[p] * src_lengths[p] <= score_stride &&
                    ref_rows[p] >= 0 && src_rows[p] >= 0 && ref_mask_offsets[p] >= 0 && src_mask_offsets[p] >= 0,
                SE3_ERR_INVALID_ARG, "superpoint_scores_stack: pair %d descriptor", p);
    T.ref_row[p] = (int)ref_rows[p];
    T.src_row[p] = (int)src_rows[p];
    T.N[p] = (int)ref_lengths[p];
    T.M[p] = (int)src_lengths[p];
    T.ref_mask[p] = (int)ref_mask_offsets[p];
    T.src_mask[p] = (int)src_mask_offsets[p];
    T.row0[p] = (int)rows;
    T.col0[p] = (int)cols;
    rows += ref_lengths[p];
    cols += src_lengths[p];
    max_m = T.M[p] > max_m ? T.M[p] : max_m;
  }
  T.row0[num_pairs] = (int)rows;
  T.col0[num_pairs] = (int)cols;
  SE3_REQUIRE(score_stride < (1ll << 31), SE3_ERR_UNSUPPORTED, "superpoint_scores_stack: %lld scores per pair", (long long)score_stride);
  hipStream_t st = (hipStream_t)stream;
  float* rowsum = workspace;               // rows floats
  float* colsum = workspace + rows;        // cols floats
  int max_n = 0;
  for (int p = 0; p < num_pairs; p++) max_n = T.N[p] > max_n ? T.N[p] : max_n;
  if (C <= 1024)
    sp_scores_stack_kernel<8><<<dim3((unsigned)((max_n + 7) / 8), (unsigned)num_pairs), 256, (size_t)8 * C * sizeof(float), st>>>(
        feats, node_masks, T, C, score_stride, scores, rowsum);
  else
    sp_scores_stack_kernel<1><<<dim3((unsigned)max_n, (unsigned)num_pairs), 256, (size_t)C * sizeof(float), st>>>(feats, node_masks, T, C,
                                                                                                                score_stride, scores, rowsum);
  if (dual_normalization)
    sp_colsum_stack_kernel<<<dim3((unsigned)((max_m + 63) / 64), (unsigned)num_pairs), 256, 0, st>>>(scores, node_masks, T,
                                                                                                  score_stride, colsum);
  sp_normalize_stack_kernel<<<dim3((unsigned)((score_stride + 255) / 256), (unsigned)num_pairs), 256, 0, st>>>(
      scores, node_masks, T, score_stride, rowsum, colsum, dual_normalization ? 1 : 0);
  SE3_CHECK_LAUNCH("superpoint_scores_stack");
  return SE3_OK;
}

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


// ---- E4 (first half): fine matching scores of all patch pairs, gathers fused -------------------------------------------------------------
// experiments/se3ete.3dmatch/model.py:186-203: ref_node_corr_knn_feats = index_select(padded fine features, ref_node_corr_knn_indices) (a
// padding index selects a zero row), the same for src, matching_scores = einsum('bnd,bmd->bnm') / sqrt(C).  One workgroup per patch pair:
// the K (= 64 or 128) rows of both patches are gathered straight into LDS in 64-channel slices (no (B, K, C) copies in HBM: 0.27 GB written
// and read back per 8-pair step before) and multiplied on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products, as the
// library GEMM it replaces).
namespace {
using f32x4m = __attribute__((ext_vector_type(4))) float;
constexpr int kPSlice = 64;                     // channels per LDS slice
constexpr int kPRow = kPSlice + 4;              // floats per staged row (+4: the 16 rows of an MFMA operand read hit different banks)

template <int KP>                               // points per patch (64: 3DMatch configurations, 128: KITTI)
__global__ __launch_bounds__(256) void patch_scores_kernel(const float* __restrict__ feats, const int64_t* __restrict__ ref_idx,
                                                           const int64_t* __restrict__ src_idx, int64_t n_rows, int C, float scale,
                                                           float* __restrict__ out) {
  __shared__ __align__(16) float sa[KP * kPRow], sb[KP * kPRow];
  __shared__ long long rows_a[KP], rows_b[KP];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * KP; i += 256) {
    const int64_t r = i < KP ? ref_idx[b * KP + i] : src_idx[b * KP + i - KP];
    (i < KP ? rows_a : rows_b)[i < KP ? i : i - KP] = (r >= 0 && r < n_rows) ? r : -1;
  }
  __syncthreads();
  // wave w owns output rows [w KP / 4, (w + 1) KP / 4) x all KP columns in 16 x 16 tiles
  constexpr int RT = KP / 64, CTN = KP / 16;    // row tiles per wave, column tiles
  f32x4m acc[RT][CTN];
#pragma unroll
  for (int r = 0; r < RT; r++)
#pragma unroll
    for (int c = 0; c < CTN; c++) acc[r][c] = f32x4m{0.f, 0.f, 0.f, 0.f};
  const int m16 = lane & 15, kq = lane >> 4;
  for (int c0 = 0; c0 < C; c0 += kPSlice) {
    // stage: thread -> (row, float4 of the slice); 16 float4 per row
    for (int i = tid; i < 2 * KP * (kPSlice / 4); i += 256) {
      const int which = i / (KP * (kPSlice / 4)), rem = i - which * (KP * (kPSlice / 4)), row = rem / (kPSlice / 4), q = rem - row * (kPSlice / 4);
      const long long src = (which ? rows_b : rows_a)[row];
      const float4 v = src >= 0 ? *reinterpret_cast<const float4*>(feats + src * C + c0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>((which ? sb : sa) + row * kPRow + 4 * q) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int k4 = 0; k4 < kPSlice / 4; k4++) {
      float av[RT], bv[CTN];
#pragma unroll
      for (int r = 0; r < RT; r++) av[r] = sa[(wave * (KP / 4) + r * 16 + m16) * kPRow + 4 * k4 + kq];
#pragma unroll
      for (int c = 0; c < CTN; c++) bv[c] = sb[(c * 16 + m16) * kPRow + 4 * k4 + kq];
#pragma unroll
      for (int r = 0; r < RT; r++)
#pragma unroll
        for (int c = 0; c < CTN; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[c], acc[r][c], 0, 0, 0);
    }
    __syncthreads();
  }
  // acc[r][c][j]: row wave KP / 4 + 16 r + 4 kq + j, column 16 c + m16
  float* o = out + b * KP * KP;
#pragma unroll
  for (int r = 0; r < RT; r++)
#pragma unroll
    for (int c = 0; c < CTN; c++)
#pragma unroll
      for (int j = 0; j < 4; j++) o[(wave * (KP / 4) + r * 16 + 4 * kq + j) * KP + c * 16 + m16] = acc[r][c][j] * scale;
}
}  // namespace

extern "C" int se3_patch_scores(const float* feats, const int64_t* ref_idx, const int64_t* src_idx, int64_t num_patches, int patch_points,
                                int64_t num_rows, int C, float scale, float* out, void* stream) {
  SE3_REQUIRE(feats && ref_idx && src_idx && out, SE3_ERR_INVALID_ARG, "patch_scores: null pointer");
  SE3_REQUIRE((patch_points == 64 || patch_points == 128) && C >= 64 && C % 64 == 0 && (reinterpret_cast<uintptr_t>(feats) & 15) == 0,
              SE3_ERR_UNSUPPORTED, "patch_scores: %d points per patch (64 or 128), %d channels (a multiple of 64)", patch_points, C);
  if (num_patches == 0) return SE3_OK;
  if (patch_points == 64) patch_scores_kernel<64><<<(unsigned)num_patches, 256, 0, (hipStream_t)stream>>>(feats, ref_idx, src_idx, num_rows, C, scale, out);
  else patch_scores_kernel<128><<<(unsigned)num_patches, 256, 0, (hipStream_t)stream>>>(feats, ref_idx, src_idx, num_rows, C, scale, out);
  SE3_CHECK_LAUNCH("patch_scores");
  return SE3_OK;
}
