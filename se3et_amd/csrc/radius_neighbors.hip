// A2: stack-mode radius neighbour search on gfx950.
//
// Replaces the nanoflann KD-tree search of the reference
// (geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91).  The result contract is integer
// and bit-exact: d2 = (dx*dx + dy*dy) + dz*dz in unfused float32 (nanoflann L2_Simple_Adaptor,
// extra/nanoflann/nanoflann.hpp:249-253), strict d2 < r*r, ascending d2 (ties broken by index; the reference leaves
// ties to std::sort), padding with the total support count.
//
// Mapping: one 64-lane wavefront owns QPW queries and keeps, per query, the `limit` (<= 64) best (d2, index) keys as a
// sorted list with ONE ENTRY PER LANE (a 64-bit key: float bits of d2 in the high word, index in the low word, so
// unsigned order = (d2, index) order).  The support cloud is streamed through LDS in SoA tiles shared by the 4 waves of
// a workgroup; every lane tests one support point per step, hits are inserted with a ballot + shuffle-shift.  The work
// is LDS/VALU bound (each support point is read from HBM once per workgroup and sits in L2 for the others).
#define SE3_EXACT_FP 1
#include "common.h"

namespace {

constexpr int kTile = 1024;        // support points per LDS tile
constexpr int kWaves = 4;          // waves per workgroup
constexpr int kQPW = 4;            // queries per wave
constexpr int kQPB = kWaves * kQPW;

struct BatchTable {
  int64_t q_start[SE3_MAX_BATCH];
  int64_t q_count[SE3_MAX_BATCH];
  int64_t s_start[SE3_MAX_BATCH];
  int64_t s_count[SE3_MAX_BATCH];
};

__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int lane) {
  int lo = __shfl_up((int)(unsigned)(v & 0xffffffffull), 1);
  int hi = __shfl_up((int)(unsigned)(v >> 32), 1);
  (void)lane;
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

__global__ __launch_bounds__(kWaves* SE3_WAVE) void radius_search_kernel(
    const float* __restrict__ q, const float* __restrict__ s, BatchTable bt, int64_t ns_total, float r2, int limit,
    int64_t* __restrict__ out, int32_t* __restrict__ max_count) {
  __shared__ float sx[kTile], sy[kTile], sz[kTile];
  const int b = blockIdx.y;
  const int64_t qn = bt.q_count[b], sn = bt.s_count[b];
  const int64_t q0 = bt.q_start[b], s0 = bt.s_start[b];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t qbase = (int64_t)blockIdx.x * kQPB + wave * kQPW;
  if ((int64_t)blockIdx.x * kQPB >= qn) return;

  float qx[kQPW], qy[kQPW], qz[kQPW];
  unsigned long long best[kQPW];
  int count[kQPW];
#pragma unroll
  for (int j = 0; j < kQPW; j++) {
    int64_t qi = qbase + j;
    bool ok = qi < qn;
    int64_t g = q0 + (ok ? qi : 0);
    qx[j] = q[3 * g + 0];
    qy[j] = q[3 * g + 1];
    qz[j] = q[3 * g + 2];
    best[j] = ~0ull;
    count[j] = 0;
  }

  for (int64_t t0 = 0; t0 < sn; t0 += kTile) {
    const int tn = (int)((sn - t0) < kTile ? (sn - t0) : kTile);
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * tn; e += kWaves * SE3_WAVE) {
      float v = s[3 * (s0 + t0) + e];
      int p = e / 3, c = e - 3 * p;
      (c == 0 ? sx : (c == 1 ? sy : sz))[p] = v;
    }
    __syncthreads();
    for (int base = 0; base < tn; base += SE3_WAVE) {
      const int p = base + lane;
      const bool valid = p < tn;
      const float px = valid ? sx[p] : 0.f, py = valid ? sy[p] : 0.f, pz = valid ? sz[p] : 0.f;
#pragma unroll
      for (int j = 0; j < kQPW; j++) {
        // unfused float32, nanoflann association order: ((dx*dx) + dy*dy) + dz*dz with d = query - support
        const float dx = __fsub_rn(qx[j], px), dy = __fsub_rn(qy[j], py), dz = __fsub_rn(qz[j], pz);
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const bool hit = valid && (d2 < r2) && (qbase + j < qn);
        unsigned long long m = __ballot(hit);
        if (m == 0ull) continue;
        count[j] += __popcll(m);
        const unsigned long long mykey =
            ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)(s0 + t0 + p - 0);
        while (m) {
          const int src = __ffsll((long long)m) - 1;
          m &= m - 1;
          const unsigned lo = __shfl((int)(unsigned)(mykey & 0xffffffffull), src);
          const unsigned hi = __shfl((int)(unsigned)(mykey >> 32), src);
          const unsigned long long cand = ((unsigned long long)hi << 32) | lo;
          const int pos = __popcll(__ballot(best[j] < cand));   // entries that stay in front of the candidate
          const unsigned long long up = shfl_up_u64(best[j], lane);
          if (lane == pos) best[j] = cand;
          else if (lane > pos) best[j] = up;
        }
      }
    }
  }

#pragma unroll
  for (int j = 0; j < kQPW; j++) {
    const int64_t qi = qbase + j;
    if (qi >= qn) continue;
    if (lane < limit) {
      const bool have = best[j] != ~0ull;
      out[(q0 + qi) * limit + lane] = have ? (int64_t)(unsigned)(best[j] & 0xffffffffull) : ns_total;
    }
    if (lane == 0) atomicMax(max_count, count[j]);
  }
}

}  // namespace

extern "C" int se3_radius_neighbors(const float* q_points, int64_t nq, const float* s_points, int64_t ns,
                                    const int64_t* q_lengths_host, const int64_t* s_lengths_host, int batch,
                                    float radius, int limit, int64_t* neighbors, int32_t* max_count, void* stream) {
  SE3_REQUIRE(batch >= 1 && batch <= SE3_MAX_BATCH, SE3_ERR_INVALID_ARG, "radius_neighbors: batch %d not in [1,%d]",
              batch, SE3_MAX_BATCH);
  SE3_REQUIRE(limit >= 1 && limit <= SE3_MAX_NEIGHBOR_LIMIT, SE3_ERR_UNSUPPORTED,
              "radius_neighbors: limit %d not in [1,%d]", limit, SE3_MAX_NEIGHBOR_LIMIT);
  SE3_REQUIRE(ns < (1ll << 31), SE3_ERR_UNSUPPORTED, "radius_neighbors: support size %lld too large", (long long)ns);
  SE3_REQUIRE(q_points && s_points && neighbors && max_count && q_lengths_host && s_lengths_host,
              SE3_ERR_INVALID_ARG, "radius_neighbors: null pointer");
  BatchTable bt;
  int64_t qs = 0, ss = 0, qmax = 0;
  for (int b = 0; b < batch; b++) {
    SE3_REQUIRE(q_lengths_host[b] >= 0 && s_lengths_host[b] >= 0, SE3_ERR_INVALID_ARG, "radius_neighbors: negative length");
    bt.q_start[b] = qs; bt.q_count[b] = q_lengths_host[b];
    bt.s_start[b] = ss; bt.s_count[b] = s_lengths_host[b];
    qs += q_lengths_host[b]; ss += s_lengths_host[b];
    if (q_lengths_host[b] > qmax) qmax = q_lengths_host[b];
  }
  SE3_REQUIRE(qs == nq && ss == ns, SE3_ERR_INVALID_ARG, "radius_neighbors: lengths sum (%lld,%lld) != sizes (%lld,%lld)",
              (long long)qs, (long long)ss, (long long)nq, (long long)ns);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(max_count, 0, sizeof(int32_t), st) != hipSuccess) {
    se3_set_error("radius_neighbors: memset failed");
    return SE3_ERR_LAUNCH;
  }
  if (nq == 0) return SE3_OK;
  dim3 grid((unsigned)se3_cdiv(qmax, kQPB), (unsigned)batch);
  radius_search_kernel<<<grid, kWaves * SE3_WAVE, 0, st>>>(q_points, s_points, bt, ns, radius * radius, limit, neighbors,
                                                         max_count);
  SE3_CHECK_LAUNCH("radius_neighbors");
  return SE3_OK;
}
