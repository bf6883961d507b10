// F1: weighted Procrustes (Kabsch) for local-to-global registration, fully on the device.
//
// Reference: geotransformer/modules/registration/procrustes.py:6-73 (weights normalised by sum + eps, weighted centroids,
// H = sum w (s - sc)(r - rc)^T, SVD on the **CPU** with a host round trip, R = V diag(1, 1, det(V U^T)) U^T) driven by
// geotransformer/modules/geotransformer/local_global_registration.py:139-194 (one hypothesis per patch pair from a host-side
// chunk loop, inlier voting, re-weighted refinement).  Here one workgroup per problem does the two weighted reductions and a
// float64 Jacobi SVD of the 3x3 covariance; problems are segments of a stacked correspondence list (segment s =
// [offsets[s], offsets[s+1])).  The refinement weights w_i = score_i * [ |r_i - T s_i| < radius ] are evaluated on the fly from
// the previous transform, so a refinement step is ONE launch.
#include "common.h"

namespace {

__device__ void jacobi_eig3(double A[3][3], double V[3][3]) {     // symmetric A -> eigenvalues on the diagonal, vectors in V columns
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; sweep++) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-40 + 1e-32 * diag) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {          // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {          // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

// R = V diag(1, 1, det(V U^T)) U^T for H = U S V^T; t = rc - R sc; writes a row-major 4x4
__device__ void kabsch(const double H[3][3], const double sc[3], const double rc[3], float* __restrict__ T) {
  double A[3][3], V[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = H[0][i] * H[0][j] + H[1][i] * H[1][j] + H[2][i] * H[2][j];   // H^T H
  jacobi_eig3(A, V);
  int o[3] = {0, 1, 2};                                   // eigenvalues in descending order
  for (int i = 0; i < 2; i++)
    for (int j = i + 1; j < 3; j++)
      if (A[o[j]][o[j]] > A[o[i]][o[i]]) { int t = o[i]; o[i] = o[j]; o[j] = t; }
  double v[3][3], u[3][3];                                // v[k], u[k]: k-th right / left singular vector
  for (int k = 0; k < 3; k++)
    for (int i = 0; i < 3; i++) v[k][i] = V[i][o[k]];
  for (int k = 0; k < 2; k++) {
    double n2 = 0;
    for (int i = 0; i < 3; i++) {
      u[k][i] = H[i][0] * v[k][0] + H[i][1] * v[k][1] + H[i][2] * v[k][2];
      n2 += u[k][i] * u[k][i];
    }
    if (k == 1) {                                         // re-orthogonalise against u0 (near-degenerate second value)
      const double d = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
      n2 = 0;
      for (int i = 0; i < 3; i++) { u[1][i] -= d * u[0][i]; n2 += u[1][i] * u[1][i]; }
    }
    const double inv = n2 > 1e-300 ? 1.0 / sqrt(n2) : 0.0;
    for (int i = 0; i < 3; i++) u[k][i] *= inv;
    if (n2 <= 1e-300) {                                   // rank < k+1: any unit vector orthogonal to the previous ones
      const double* b = u[0];
      double e[3] = {fabs(b[0]) < 0.9 ? 1.0 : 0.0, fabs(b[0]) < 0.9 ? 0.0 : 1.0, 0.0};
      if (k == 0) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; }
      else {
        const double d = e[0] * b[0] + e[1] * b[1] + e[2] * b[2];
        double m2 = 0;
        for (int i = 0; i < 3; i++) { u[1][i] = e[i] - d * b[i]; m2 += u[1][i] * u[1][i]; }
        for (int i = 0; i < 3; i++) u[1][i] /= sqrt(m2);
      }
    }
  }
  // third vectors by cross products: det([u0 u1 u2]) = det([v0 v1 v2']) = +1 with v2' = v0 x v1, i.e. the reflection fix
  // diag(1, 1, det(V U^T)) is already applied
  u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
  u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
  u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  double w2[3] = {v[0][1] * v[1][2] - v[0][2] * v[1][1], v[0][2] * v[1][0] - v[0][0] * v[1][2],
                  v[0][0] * v[1][1] - v[0][1] * v[1][0]};
  double R[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i][j] = v[0][i] * u[0][j] + v[1][i] * u[1][j] + w2[i] * u[2][j];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[4 * i + j] = (float)R[i][j];
    T[4 * i + 3] = (float)(rc[i] - (R[i][0] * sc[0] + R[i][1] * sc[1] + R[i][2] * sc[2]));
  }
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

__device__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) r += sh[w];
  return r;
}

// one workgroup per segment
__global__ __launch_bounds__(256) void procrustes_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                         const float* __restrict__ score, const int64_t* __restrict__ offsets,
                                                         const float* __restrict__ T_prev, int gate_stride, float radius,
                                                         float eps, float* __restrict__ T_out) {
  __shared__ double sh[4];
  const int64_t b0 = offsets[blockIdx.x], b1 = offsets[blockIdx.x + 1];
  float Tp[12];
  const bool gate = T_prev != nullptr;
  if (gate)
    for (int i = 0; i < 12; i++) Tp[i] = T_prev[(size_t)blockIdx.x * gate_stride + i];     // gate_stride 0: one transform for all
  auto weight = [&](int64_t i) -> float {
    float w = score[i];
    w = w < 0.f ? 0.f : w;
    if (gate) {
      const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
      const float dx = ref[3 * i] - (Tp[0] * sx + Tp[1] * sy + Tp[2] * sz + Tp[3]);
      const float dy = ref[3 * i + 1] - (Tp[4] * sx + Tp[5] * sy + Tp[6] * sz + Tp[7]);
      const float dz = ref[3 * i + 2] - (Tp[8] * sx + Tp[9] * sy + Tp[10] * sz + Tp[11]);
      if (!(sqrtf(dx * dx + dy * dy + dz * dz) < radius)) w = 0.f;
    }
    return w;
  };
  // one pass: raw weighted moments in float64 (sum w, sum w s, sum w r, sum w s r^T), centred afterwards -- in double the
  // cancellation of the centring is harmless (coordinates ~1e0..1e2, 53-bit sums)
  double m[16];
  for (int k = 0; k < 16; k++) m[k] = 0;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) {
    const double w = (double)weight(i);
    const double s3[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]}, r3[3] = {ref[3 * i], ref[3 * i + 1], ref[3 * i + 2]};
    m[0] += w;
    for (int d = 0; d < 3; d++) { m[1 + d] += w * s3[d]; m[4 + d] += w * r3[d]; }
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 3; c++) m[7 + 3 * a + c] += w * s3[a] * r3[c];
  }
  __shared__ double red[4][16];
  for (int k = 0; k < 16; k++) {
    double v = m[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  (void)sh;
  if (threadIdx.x != 0) return;
  for (int k = 0; k < 16; k++) m[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  const double inv = 1.0 / (m[0] + (double)eps);       // weights are normalised by (sum + eps) as in the reference
  const double W1 = m[0] * inv;                        // sum of the normalised weights (slightly below 1)
  double sc[3], rc[3], H[3][3];
  for (int d = 0; d < 3; d++) { sc[d] = m[1 + d] * inv; rc[d] = m[4 + d] * inv; }
  // H = sum w' (s - sc)(r - rc)^T = sum w' s r^T - sc (sum w' r)^T - (sum w' s) rc^T + W1 sc rc^T = M - (2 - W1) sc rc^T
  for (int a = 0; a < 3; a++)
    for (int c = 0; c < 3; c++) H[a][c] = m[7 + 3 * a + c] * inv - (2.0 - W1) * sc[a] * rc[c];
  kabsch(H, sc, rc, T_out + 16 * blockIdx.x);
}

// votes[b] = number of correspondences with |ref - T_b src| < radius ; inlier mask optional (for the chosen hypothesis)
__global__ __launch_bounds__(256) void vote_kernel(const float* __restrict__ src, const float* __restrict__ ref, int64_t total,
                                                   const float* __restrict__ T, const int64_t* __restrict__ range_begin,
                                                   const int64_t* __restrict__ range_end, float radius,
                                                   int32_t* __restrict__ votes) {
  __shared__ int shv[4];
  const int64_t lo = range_begin ? range_begin[blockIdx.x] : 0, hi = range_end ? range_end[blockIdx.x] : total;
  const float* Tb = T + 16 * blockIdx.x;
  float Tp[12];
  for (int i = 0; i < 12; i++) Tp[i] = Tb[i];
  int cnt = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
    const float dx = ref[3 * i] - (Tp[0] * sx + Tp[1] * sy + Tp[2] * sz + Tp[3]);
    const float dy = ref[3 * i + 1] - (Tp[4] * sx + Tp[5] * sy + Tp[6] * sz + Tp[7]);
    const float dz = ref[3 * i + 2] - (Tp[8] * sx + Tp[9] * sy + Tp[10] * sz + Tp[11]);
    cnt += (sqrtf(dx * dx + dy * dy + dz * dz) < radius) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0) shv[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) votes[blockIdx.x] = shv[0] + shv[1] + shv[2] + shv[3];
}

// Mutual top-k correspondence mask of local_global_registration.py:104-131: entry (i, j) of patch pair b survives when it is among
// the k largest of its row AND of its column (ties by index), both above the confidence threshold, and both points are valid.
// One workgroup per patch pair, the (rows, cols) score matrix in LDS; replaces two torch.topk + two scatters per call.
__global__ __launch_bounds__(256) void mutual_topk_kernel(const float* __restrict__ scores, const unsigned char* __restrict__ row_masks,
                                                          const unsigned char* __restrict__ col_masks, int R, int C, int k,
                                                          float threshold, unsigned char* __restrict__ out) {
  extern __shared__ float sm[];              // R * C scores
  const int b = blockIdx.x;
  const float* S = scores + (size_t)b * R * C;
  for (int i = threadIdx.x; i < R * C; i += 256) sm[i] = S[i];
  __syncthreads();
  for (int e = threadIdx.x; e < R * C; e += 256) {
    const int i = e / C, j = e - i * C;
    const float v = sm[e];
    bool keep = v > threshold && row_masks[(size_t)b * R + i] && col_masks[(size_t)b * C + j];
    if (keep) {
      int ahead = 0;                         // entries of row i that precede (i, j) in a descending, index-stable order
      for (int jj = 0; jj < C; jj++) {
        const float o = sm[i * C + jj];
        ahead += (o > v) || (o == v && jj < j);
      }
      keep = ahead < k;
    }
    if (keep) {
      int ahead = 0;
      for (int ii = 0; ii < R; ii++) {
        const float o = sm[ii * C + j];
        ahead += (o > v) || (o == v && ii < i);
      }
      keep = ahead < k;
    }
    out[(size_t)b * R * C + e] = keep ? 1 : 0;
  }
}

}  // namespace

extern "C" int se3_weighted_procrustes_segments(const float* src_points, const float* ref_points, const float* scores,
                                                const int64_t* segment_offsets, int num_segments, const float* gate_transforms,
                                                int gate_per_segment, float gate_radius, float eps, float* transforms,
                                                void* stream) {
  SE3_REQUIRE(src_points && ref_points && scores && segment_offsets && transforms, SE3_ERR_INVALID_ARG,
              "weighted_procrustes: null pointer");
  SE3_REQUIRE(num_segments >= 0, SE3_ERR_INVALID_ARG, "weighted_procrustes: negative segment count");
  if (num_segments == 0) return SE3_OK;
  procrustes_kernel<<<num_segments, 256, 0, (hipStream_t)stream>>>(src_points, ref_points, scores, segment_offsets,
                                                                   gate_transforms, gate_per_segment ? 16 : 0, gate_radius, eps,
                                                                   transforms);
  SE3_CHECK_LAUNCH("weighted_procrustes");
  return SE3_OK;
}

extern "C" int se3_weighted_procrustes(const float* src_points, const float* ref_points, const float* scores,
                                       const int64_t* segment_offsets, int num_segments, const float* gate_transform,
                                       float gate_radius, float eps, float* transforms, void* stream) {
  return se3_weighted_procrustes_segments(src_points, ref_points, scores, segment_offsets, num_segments, gate_transform, 0,
                                          gate_radius, eps, transforms, stream);
}

extern "C" int se3_count_inliers_ranges(const float* src_points, const float* ref_points, int64_t num_points,
                                        const float* transforms, int num_transforms, const int64_t* range_begin,
                                        const int64_t* range_end, float radius, int32_t* votes, void* stream) {
  SE3_REQUIRE(src_points && ref_points && transforms && votes, SE3_ERR_INVALID_ARG, "count_inliers: null pointer");
  SE3_REQUIRE((range_begin == nullptr) == (range_end == nullptr), SE3_ERR_INVALID_ARG, "count_inliers: ranges go together");
  if (num_transforms <= 0) return SE3_OK;
  vote_kernel<<<num_transforms, 256, 0, (hipStream_t)stream>>>(src_points, ref_points, num_points, transforms, range_begin,
                                                              range_end, radius, votes);
  SE3_CHECK_LAUNCH("count_inliers");
  return SE3_OK;
}

extern "C" int se3_count_inliers(const float* src_points, const float* ref_points, int64_t num_points,
                                 const float* transforms, int num_transforms, float radius, int32_t* votes, void* stream) {
  return se3_count_inliers_ranges(src_points, ref_points, num_points, transforms, num_transforms, nullptr, nullptr, radius, votes,
                                  stream);
}

extern "C" int se3_mutual_topk_mask(const float* scores, const uint8_t* row_masks, const uint8_t* col_masks, int batch, int rows,
                                    int cols, int k, float threshold, uint8_t* mask, void* stream) {
  SE3_REQUIRE(scores && row_masks && col_masks && mask, SE3_ERR_INVALID_ARG, "mutual_topk_mask: null pointer");
  SE3_REQUIRE(batch >= 0 && rows >= 1 && cols >= 1 && k >= 1 && (size_t)rows * cols * sizeof(float) <= 64 * 1024, SE3_ERR_UNSUPPORTED,
              "mutual_topk_mask: batch %d rows %d cols %d k %d (rows * cols <= 16384)", batch, rows, cols, k);
  if (batch == 0) return SE3_OK;
  mutual_topk_kernel<<<(unsigned)batch, 256, (size_t)rows * cols * sizeof(float), (hipStream_t)stream>>>(
      scores, row_masks, col_masks, rows, cols, k, threshold, mask);
  SE3_CHECK_LAUNCH("mutual_topk_mask");
  return SE3_OK;
}

