// E3 / G1 helpers: nearest-neighbour selections on the superpoint level, one launch each instead of a distance matrix + torch.topk.
//
//   se3_knn3                     the 3 nearest OTHER points of every point of a cloud (GeometricStructureEmbedding.get_embedding_indices,
//                                geotransformer/modules/geotransformer/geotransformer.py:69-90: topk(k+1) of the distance map, first column
//                                dropped)
//   se3_point_to_node_partition  point_to_node_partition (geotransformer/modules/ops/pointcloud_partition.py:60-107): every fine point
//                                goes to its nearest node; every node lists the `limit` nearest of ITS OWN points, padded with N
//
// Distances follow pairwise_distance (modules/ops/pairwise_distance.py:4-30): (|x|^2 - 2 x.y) + |y|^2, clamped at 0; ties are
// ordered by index.  One wave per query row, candidates strided over the lanes, sorted lists kept one entry per lane.
#include "common.h"

namespace {

__device__ __forceinline__ float sq_norm(float x, float y, float z) { return se3_ref_sq_norm(x, y, z); }
__device__ __forceinline__ float pair_dist(float qx, float qy, float qz, float q2, float sx, float sy, float sz, float s2) {
  return se3_ref_sq_dist(qx, qy, qz, q2, sx, sy, sz, s2);          // (common.h: the reference's float32 expression, operation by operation)
}
__device__ __forceinline__ unsigned long long make_key(float d, unsigned idx) {
  return ((unsigned long long)__float_as_uint(d) << 32) | idx;            // d >= 0: float bits order like the values
}
__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) {
  const unsigned lo = __shfl((unsigned)(v & 0xffffffffull), src), hi = __shfl((unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_up1(unsigned long long v) {
  const unsigned lo = __shfl_up((unsigned)(v & 0xffffffffull), 1), hi = __shfl_up((unsigned)(v >> 32), 1);
  return ((unsigned long long)hi << 32) | lo;
}
// insert `cand` (valid on the lanes where has != 0) one at a time into the wave-wide ascending list `best` (lane = rank)
__device__ __forceinline__ void wave_insert(unsigned long long& best, unsigned long long cand, bool has) {
  const int lane = threadIdx.x & 63;
  unsigned long long pending = __ballot(has);
  while (pending) {
    const int src = __ffsll((long long)pending) - 1;
    pending &= pending - 1;
    const unsigned long long c = shfl64(cand, src);
    const int pos = __popcll(__ballot(best < c));
    const unsigned long long up = shfl_up1(best);
    if (lane == pos) best = c;
    else if (lane > pos) best = up;
  }
}

__global__ __launch_bounds__(256) void knn3_kernel(const float* __restrict__ p, int N, int64_t* __restrict__ knn) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const float qx = p[3 * i], qy = p[3 * i + 1], qz = p[3 * i + 2], q2 = sq_norm(qx, qy, qz);
  unsigned long long best = ~0ull;          // lanes 0..3 hold the 4 smallest (distance, index) keys
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    unsigned long long c = ~0ull;
    if (j < N) {
      const float sx = p[3 * j], sy = p[3 * j + 1], sz = p[3 * j + 2];
      c = make_key(pair_dist(qx, qy, qz, q2, sx, sy, sz, sq_norm(sx, sy, sz)), (unsigned)j);
    }
    const unsigned long long worst = shfl64(best, 3);            // only the 4 smallest matter
    wave_insert(best, c, j < N && c < worst);
  }
  // the reference drops the first column of topk(k + 1) (the point itself at distance 0)
  if (lane >= 1 && lane <= 3) knn[(size_t)i * 3 + lane - 1] = best == ~0ull ? (int64_t)i : (int64_t)(unsigned)(best & 0xffffffffull);
}

__global__ __launch_bounds__(256) void nearest_node_kernel(const float* __restrict__ pts, const float* __restrict__ nodes, int N,
                                                           int M, int64_t* __restrict__ point_to_node,
                                                           unsigned char* __restrict__ node_masks) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2], p2 = sq_norm(px, py, pz);
  unsigned long long best = ~0ull;
  for (int m = lane; m < M; m += 64) {
    const float nx = nodes[3 * m], ny = nodes[3 * m + 1], nz = nodes[3 * m + 2];
    // pairwise_distance(nodes, points): x = node, y = point
    const unsigned long long c = make_key(pair_dist(nx, ny, nz, sq_norm(nx, ny, nz), px, py, pz, p2), (unsigned)m);
    best = c < best ? c : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffull), o), hi = __shfl_xor((unsigned)(best >> 32), o);
    const unsigned long long c = ((unsigned long long)hi << 32) | lo;
    best = c < best ? c : best;
  }
  if (lane == 0) {
    const unsigned m = (unsigned)(best & 0xffffffffull);
    point_to_node[i] = (int64_t)m;
    node_masks[m] = 1;
  }
}

// two-register version of wave_insert: a 128-entry ascending list, ranks 0..63 in b0, 64..127 in b1
__device__ __forceinline__ void wave_insert2(unsigned long long& b0, unsigned long long& b1, unsigned long long cand, bool has) {
  const int lane = threadIdx.x & 63;
  unsigned long long pending = __ballot(has);
  while (pending) {
    const int src = __ffsll((long long)pending) - 1;
    pending &= pending - 1;
    const unsigned long long c = shfl64(cand, src);
    const int pos = __popcll(__ballot(b0 < c)) + __popcll(__ballot(b1 < c));
    const unsigned long long up0 = shfl_up1(b0), up1 = shfl_up1(b1), carry = shfl64(b0, 63);
    if (pos < 64) {
      if (lane == pos) b0 = c;
      else if (lane > pos) b0 = up0;
      b1 = lane == 0 ? carry : up1;
    } else {
      const int p1 = pos - 64;
      if (lane == p1) b1 = c;
      else if (lane > p1) b1 = up1;
    }
  }
}

template <bool WIDE>        // WIDE: limit in (64, 128]
__global__ __launch_bounds__(256) void node_knn_kernel(const float* __restrict__ pts, const float* __restrict__ nodes,
                                                       const int64_t* __restrict__ point_to_node, int N, int M, int limit,
                                                       int64_t* __restrict__ knn, unsigned char* __restrict__ knn_masks) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float nx = nodes[3 * m], ny = nodes[3 * m + 1], nz = nodes[3 * m + 2], n2 = sq_norm(nx, ny, nz);
  unsigned long long best = ~0ull, best1 = ~0ull;          // lane r = r-th (64 + r-th) nearest own point
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    const bool own = j < N && point_to_node[j] == (int64_t)m;
    unsigned long long c = ~0ull;
    if (own) {
      const float px = pts[3 * j], py = pts[3 * j + 1], pz = pts[3 * j + 2];
      c = make_key(pair_dist(nx, ny, nz, n2, px, py, pz, sq_norm(px, py, pz)), (unsigned)j);
    }
    if (WIDE) wave_insert2(best, best1, c, own);
    else wave_insert(best, c, own);
  }
  if (lane < limit) {
    const bool have = best != ~0ull;
    knn[(size_t)m * limit + lane] = have ? (int64_t)(unsigned)(best & 0xffffffffull) : (int64_t)N;
    knn_masks[(size_t)m * limit + lane] = have ? 1 : 0;
  }
  if (WIDE && 64 + lane < limit) {
    const bool have = best1 != ~0ull;
    knn[(size_t)m * limit + 64 + lane] = have ? (int64_t)(unsigned)(best1 & 0xffffffffull) : (int64_t)N;
    knn_masks[(size_t)m * limit + 64 + lane] = have ? 1 : 0;
  }
}

// ---- stack mode: the clouds of several pairs in one launch -------------------------------------------------------------
constexpr int kMaxPartClouds = SE3_MAX_BATCH;
struct PartClouds {
  int p0[kMaxPartClouds + 1];     // first point of cloud c in the stacked point array (p0[n] = total points)
  int m0[kMaxPartClouds + 1];     // first node of cloud c in the stacked node array
  int n;
};

// all clouds of a batch in one launch: point i looks only at the points of its own cloud; LOCAL indices (0 .. N_c - 1)
__global__ __launch_bounds__(256) void knn3_stack_table_kernel(const float* __restrict__ pts, PartClouds T, int num_clouds,
                                                               int total, int64_t* __restrict__ knn) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= total) return;
  int c = 0;
  for (int k = 1; k < num_clouds; k++)
    if (i >= T.p0[k]) c = k;
  const int b0 = T.p0[c], N = T.p0[c + 1] - b0;
  const float* p = pts + 3 * (size_t)b0;
  const int li = i - b0;
  const float qx = p[3 * li], qy = p[3 * li + 1], qz = p[3 * li + 2], q2 = sq_norm(qx, qy, qz);
  unsigned long long best = ~0ull;
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    unsigned long long cand = ~0ull;
    if (j < N) {
      const float sx = p[3 * j], sy = p[3 * j + 1], sz = p[3 * j + 2];
      cand = make_key(pair_dist(qx, qy, qz, q2, sx, sy, sz, sq_norm(sx, sy, sz)), (unsigned)j);
    }
    const unsigned long long worst = shfl64(best, 3);
    wave_insert(best, cand, j < N && cand < worst);
  }
  if (lane >= 1 && lane <= 3) knn[(size_t)i * 3 + lane - 1] = best == ~0ull ? (int64_t)li : (int64_t)(unsigned)(best & 0xffffffffull);
}

// point i -> its cloud's nearest node (GLOBAL node index)
__global__ __launch_bounds__(256) void nearest_node_stack_kernel(const float* __restrict__ pts, const float* __restrict__ nodes,
                                                                 PartClouds T, int64_t* __restrict__ point_to_node,
                                                                 unsigned char* __restrict__ node_masks) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= T.p0[T.n]) return;
  int c = 0;
  for (int k = 1; k < T.n; k++)
    if (i >= T.p0[k]) c = k;
  const int mb = T.m0[c], me = T.m0[c + 1];
  const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2], p2 = sq_norm(px, py, pz);
  unsigned long long best = ~0ull;
  for (int m = mb + lane; m < me; m += 64) {
    const float nx = nodes[3 * m], ny = nodes[3 * m + 1], nz = nodes[3 * m + 2];
    const unsigned long long cand = make_key(pair_dist(nx, ny, nz, sq_norm(nx, ny, nz), px, py, pz, p2), (unsigned)m);
    best = cand < best ? cand : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffull), o), hi = __shfl_xor((unsigned)(best >> 32), o);
    const unsigned long long cand = ((unsigned long long)hi << 32) | lo;
    best = cand < best ? cand : best;
  }
  if (lane == 0) {
    const unsigned m = (unsigned)(best & 0xffffffffull);
    point_to_node[i] = (int64_t)m;
    node_masks[m] = 1;
  }
}

// node m (global) -> the `limit` nearest of its own points (GLOBAL point indices, padded with the total point count)
template <bool WIDE>
__global__ __launch_bounds__(256) void node_knn_stack_kernel(const float* __restrict__ pts, const float* __restrict__ nodes,
                                                             const int64_t* __restrict__ point_to_node, PartClouds T, int limit,
                                                             int64_t* __restrict__ knn, unsigned char* __restrict__ knn_masks) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= T.m0[T.n]) return;
  int c = 0;
  for (int k = 1; k < T.n; k++)
    if (m >= T.m0[k]) c = k;
  const int pb = T.p0[c], pe = T.p0[c + 1], total = T.p0[T.n];
  const float nx = nodes[3 * m], ny = nodes[3 * m + 1], nz = nodes[3 * m + 2], n2 = sq_norm(nx, ny, nz);
  unsigned long long best = ~0ull, best1 = ~0ull;
  for (int j0 = pb; j0 < pe; j0 += 64) {
    const int j = j0 + lane;
    const bool own = j < pe && point_to_node[j] == (int64_t)m;
    unsigned long long cand = ~0ull;
    if (own) {
      const float px = pts[3 * j], py = pts[3 * j + 1], pz = pts[3 * j + 2];
      cand = make_key(pair_dist(nx, ny, nz, n2, px, py, pz, sq_norm(px, py, pz)), (unsigned)j);
    }
    if (WIDE) wave_insert2(best, best1, cand, own);
    else wave_insert(best, cand, own);
  }
  if (lane < limit) {
    const bool have = best != ~0ull;
    knn[(size_t)m * limit + lane] = have ? (int64_t)(unsigned)(best & 0xffffffffull) : (int64_t)total;
    knn_masks[(size_t)m * limit + lane] = have ? 1 : 0;
  }
  if (WIDE && 64 + lane < limit) {
    const bool have = best1 != ~0ull;
    knn[(size_t)m * limit + 64 + lane] = have ? (int64_t)(unsigned)(best1 & 0xffffffffull) : (int64_t)total;
    knn_masks[(size_t)m * limit + 64 + lane] = have ? 1 : 0;
  }
}

}  // namespace

extern "C" int se3_point_to_node_partition_stack(const float* points, const float* nodes, const int64_t* point_lengths,
                                                 const int64_t* node_lengths, int num_clouds, int limit,
                                                 int64_t* point_to_node, uint8_t* node_masks, int64_t* node_knn_indices,
                                                 uint8_t* node_knn_masks, void* stream) {
  SE3_REQUIRE(points && nodes && point_lengths && node_lengths && point_to_node && node_masks && node_knn_indices && node_knn_masks,
              SE3_ERR_INVALID_ARG, "point_to_node_partition_stack: null pointer");
  SE3_REQUIRE(num_clouds >= 1 && num_clouds <= kMaxPartClouds && limit >= 1 && limit <= 128, SE3_ERR_UNSUPPORTED,
              "point_to_node_partition_stack: %d clouds (1..%d), limit %d (<= 128)", num_clouds, kMaxPartClouds, limit);
  PartClouds T{};
  T.n = num_clouds;
  int64_t np = 0, nm = 0;
  for (int c = 0; c < num_clouds; c++) {
    SE3_REQUIRE(point_lengths[c] >= 1 && node_lengths[c] >= 1, SE3_ERR_INVALID_ARG, "point_to_node_partition_stack: empty cloud %d", c);
    T.p0[c] = (int)np;
    T.m0[c] = (int)nm;
    np += point_lengths[c];
    nm += node_lengths[c];
  }
  SE3_REQUIRE(np < (1ll << 31) && nm < (1ll << 31), SE3_ERR_UNSUPPORTED, "point_to_node_partition_stack: too many points");
  T.p0[num_clouds] = (int)np;
  T.m0[num_clouds] = (int)nm;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(node_masks, 0, (size_t)nm, st) != hipSuccess) {
    se3_set_error("point_to_node_partition_stack: memset failed");
    return SE3_ERR_LAUNCH;
  }
  nearest_node_stack_kernel<<<(unsigned)se3_cdiv(np, 4), 256, 0, st>>>(points, nodes, T, point_to_node, node_masks);
  if (limit <= 64)
    node_knn_stack_kernel<false><<<(unsigned)se3_cdiv(nm, 4), 256, 0, st>>>(points, nodes, point_to_node, T, limit,
                                                                            node_knn_indices, node_knn_masks);
  else
    node_knn_stack_kernel<true><<<(unsigned)se3_cdiv(nm, 4), 256, 0, st>>>(points, nodes, point_to_node, T, limit,
                                                                           node_knn_indices, node_knn_masks);
  SE3_CHECK_LAUNCH("point_to_node_partition_stack");
  return SE3_OK;
}

extern "C" int se3_knn3(const float* points, int N, int64_t* knn, void* stream) {
  SE3_REQUIRE(points && knn, SE3_ERR_INVALID_ARG, "knn3: null pointer");
  SE3_REQUIRE(N >= 1, SE3_ERR_INVALID_ARG, "knn3: N %d", N);
  knn3_kernel<<<(unsigned)se3_cdiv(N, 4), 256, 0, (hipStream_t)stream>>>(points, N, knn);
  SE3_CHECK_LAUNCH("knn3");
  return SE3_OK;
}

extern "C" int se3_knn3_stack(const float* points, const int64_t* lengths, int num_clouds, int64_t* knn, void* stream) {
  SE3_REQUIRE(points && lengths && knn, SE3_ERR_INVALID_ARG, "knn3_stack: null pointer");
  SE3_REQUIRE(num_clouds >= 1 && num_clouds <= kMaxPartClouds, SE3_ERR_UNSUPPORTED, "knn3_stack: %d clouds (1..%d)", num_clouds,
              kMaxPartClouds);
  PartClouds T{};
  int64_t total = 0;
  for (int c = 0; c < num_clouds; c++) {
    SE3_REQUIRE(lengths[c] >= 1, SE3_ERR_INVALID_ARG, "knn3_stack: empty cloud %d", c);
    T.p0[c] = (int)total;
    total += lengths[c];
  }
  SE3_REQUIRE(total < (1ll << 31), SE3_ERR_UNSUPPORTED, "knn3_stack: too many points");
  T.p0[num_clouds] = (int)total;
  knn3_stack_table_kernel<<<(unsigned)se3_cdiv(total, 4), 256, 0, (hipStream_t)stream>>>(points, T, num_clouds, (int)total, knn);
  SE3_CHECK_LAUNCH("knn3_stack");
  return SE3_OK;
}

extern "C" int se3_point_to_node_partition(const float* points, const float* nodes, int N, int M, int limit,
                                           int64_t* point_to_node, uint8_t* node_masks, int64_t* node_knn_indices,
                                           uint8_t* node_knn_masks, void* stream) {
  SE3_REQUIRE(points && nodes && point_to_node && node_masks && node_knn_indices && node_knn_masks, SE3_ERR_INVALID_ARG,
              "point_to_node_partition: null pointer");
  SE3_REQUIRE(N >= 1 && M >= 1 && limit >= 1 && limit <= 128, SE3_ERR_UNSUPPORTED,
              "point_to_node_partition: N %d M %d limit %d (limit <= 128)", N, M, limit);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(node_masks, 0, (size_t)M, st) != hipSuccess) {
    se3_set_error("point_to_node_partition: memset failed");
    return SE3_ERR_LAUNCH;
  }
  nearest_node_kernel<<<(unsigned)se3_cdiv(N, 4), 256, 0, st>>>(points, nodes, N, M, point_to_node, node_masks);
  if (limit <= 64)
    node_knn_kernel<false><<<(unsigned)se3_cdiv(M, 4), 256, 0, st>>>(points, nodes, point_to_node, N, M, limit,
                                                                    node_knn_indices, node_knn_masks);
  else
    node_knn_kernel<true><<<(unsigned)se3_cdiv(M, 4), 256, 0, st>>>(points, nodes, point_to_node, N, M, limit,
                                                                   node_knn_indices, node_knn_masks);
  SE3_CHECK_LAUNCH("point_to_node_partition");
  return SE3_OK;
}
