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

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
  const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), mask), hi = __shfl_xor((unsigned)(v >> 32), mask);
  return ((unsigned long long)hi << 32) | lo;
}
// ascending bitonic sort of one 64-bit key per lane over the wave (21 compare-exchange steps)
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const unsigned long long o = shfl_xor_u64(v, j);
      const bool up = ((lane & k) == 0) || k == 64;          // direction of this lane's block (the last stage is ascending)
      const bool lower = (lane & j) == 0;
      const unsigned long long mn = v < o ? v : o, mx = v < o ? o : v;
      v = (lower == up) ? mn : mx;
    }
  }
  return v;
}
// the 64 smallest of two ascending 64-key lists, ascending
__device__ __forceinline__ unsigned long long wave_merge_lower(unsigned long long a, unsigned long long b) {
  const int lane = threadIdx.x & 63;
  const unsigned lo = __shfl((unsigned)(b & 0xffffffffull), 63 - lane), hi = __shfl((unsigned)(b >> 32), 63 - lane);
  const unsigned long long br = ((unsigned long long)hi << 32) | lo;        // b reversed: min(a, reverse(b)) is bitonic
  unsigned long long v = a < br ? a : br;
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const unsigned long long o = shfl_xor_u64(v, j);
    const bool lower = (lane & j) == 0;
    const unsigned long long mn = v < o ? v : o, mx = v < o ? o : v;
    v = lower ? mn : mx;
  }
  return v;
}

// Rows whose result depends on the ORDER of exactly tied distances (the reference leaves it to its k-d tree walk + an unstable sort,
// csrc/radius_ties.hip): two neighbouring entries of the sorted list with the same d2, the first of them inside the kept `limit` columns --
// a tie among the kept entries, or between the last kept one and the first one cut.  `best`: the wave's sorted (d2, index) list, one entry per
// lane; `count`: in-radius points of the query.  With limit = 64 the entry behind the cut is not in the list: a row with more than 64
// matches is flagged.  The row number is appended to tie_rows behind the device counter tie_count (order of arrival: rows are independent).
__device__ __forceinline__ void flag_tie_row(unsigned long long best, int count, int limit, int64_t row, int32_t* __restrict__ tie_rows,
                                             int32_t* __restrict__ tie_count) {
  const int lane = threadIdx.x & 63;
  const unsigned d_here = (unsigned)(best >> 32), d_next = (unsigned)__shfl_down((int)d_here, 1);
  const int kept = count < 64 ? count : 64;
  const bool tie = (lane + 1 < kept && lane < limit && d_here == d_next) || (limit == 64 && count > 64);
  if (__ballot(tie) != 0ull && lane == 0) tie_rows[atomicAdd(tie_count, 1)] = (int32_t)row;
}

__global__ __launch_bounds__(kWaves* SE3_WAVE) void radius_search_kernel(
    const float* __restrict__ q, const float* __restrict__ s, BatchTable bt, int64_t ns_total, float r2, int limit,
    int64_t* __restrict__ out, int32_t* __restrict__ max_count, int32_t* __restrict__ tie_rows, int32_t* __restrict__ tie_count) {
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
    // same-line atomics serialise in the L2 (~7 ns each: 80 000 queries = 0.5 ms); the running maximum is monotonic, so a plain
    // (possibly stale) read filters almost all of them
    if (lane == 0 && count[j] > __atomic_load_n(max_count + b, __ATOMIC_RELAXED)) atomicMax(max_count + b, count[j]);
    if (tie_rows != nullptr) flag_tie_row(best[j], count[j], limit, q0 + qi, tie_rows, tie_count);
  }
}

}  // namespace

extern "C" int se3_radius_neighbors_ties(const float* q_points, int64_t nq, const float* s_points, int64_t ns,
                                         const int64_t* q_lengths_host, const int64_t* s_lengths_host, int batch,
                                         float radius, int limit, int64_t* neighbors, int32_t* max_count, int32_t* tie_rows,
                                         int32_t* tie_count, void* stream) {
  SE3_REQUIRE((tie_rows == nullptr) == (tie_count == nullptr), SE3_ERR_INVALID_ARG, "radius_neighbors: tie_rows and tie_count go together");
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
  if (hipMemsetAsync(max_count, 0, sizeof(int32_t) * batch, st) != hipSuccess) {
    se3_set_error("radius_neighbors: memset failed");
    return SE3_ERR_LAUNCH;
  }
  if (nq == 0) return SE3_OK;
  dim3 grid((unsigned)se3_cdiv(qmax, kQPB), (unsigned)batch);
  radius_search_kernel<<<grid, kWaves * SE3_WAVE, 0, st>>>(q_points, s_points, bt, ns, radius * radius, limit, neighbors,
                                                         max_count, tie_rows, tie_count);
  SE3_CHECK_LAUNCH("radius_neighbors");
  return SE3_OK;
}

extern "C" int se3_radius_neighbors(const float* q_points, int64_t nq, const float* s_points, int64_t ns,
                                    const int64_t* q_lengths_host, const int64_t* s_lengths_host, int batch,
                                    float radius, int limit, int64_t* neighbors, int32_t* max_count, void* stream) {
  return se3_radius_neighbors_ties(q_points, nq, s_points, ns, q_lengths_host, s_lengths_host, batch, radius, limit, neighbors, max_count,
                                   nullptr, nullptr, stream);
}

// =====================================================================================================================
// Uniform-grid variant for large supports.  The support cloud is binned once into cells of edge >= radius (counting sort on
// the device: cell histogram -> prefix sum -> scatter of (x, y, z, index)); a query then only visits the 3 x 3 x 3 block of
// cells around it, i.e. nine contiguous runs of the cell-sorted array.  The distance arithmetic, the strict d2 < r2 test and
// the (d2, index) ranking are exactly those of the exhaustive kernel above, so the results are bit-identical; only the
// candidates that cannot be within the radius are skipped.  One grid serves every search that shares support and radius
// (stage neighbours, sub-sampling, and the previous stage's up-sampling).
// =====================================================================================================================
namespace {

constexpr int kGridCap = 64;                       // cells per axis at most
constexpr int kCellCap = kGridCap * kGridCap * kGridCap;

struct GridMeta {
  float org[3];
  float inv_cell;
  int dim[3];
  int ncells;
};

struct GridLayout {
  GridMeta* meta;        // [batch]
  int* cell_start;       // [batch][kCellCap + 1]
  int* cell_fill;        // [batch][kCellCap]
  int* cell_of;          // [ns]
  float4* sorted;        // [ns]  (x, y, z, bits of the stacked support index)
};

size_t grid_carve(int64_t ns, int batch, char* base, GridLayout* L) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    char* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  };
  GridLayout l;
  l.meta = (GridMeta*)take(sizeof(GridMeta) * batch);
  l.cell_start = (int*)take(sizeof(int) * (size_t)batch * (kCellCap + 1));
  l.cell_fill = (int*)take(sizeof(int) * (size_t)batch * kCellCap);
  l.cell_of = (int*)take(sizeof(int) * (size_t)(ns > 0 ? ns : 1));
  l.sorted = (float4*)take(sizeof(float4) * (size_t)(ns > 0 ? ns : 1));
  if (L) *L = l;
  return (off + 255) & ~(size_t)255;
}

__device__ __forceinline__ int cell_coord(float v, float org, float inv_cell, int dim) {
  int c = (int)floorf((v - org) * inv_cell);
  return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

__global__ __launch_bounds__(1024) void grid_bounds_kernel(const float* __restrict__ s, BatchTable bt, float radius,
                                                           GridLayout G) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  const int64_t n = bt.s_count[b];
  const float* p = s + 3 * bt.s_start[b];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = threadIdx.x; i < n; i += 1024)
    for (int d = 0; d < 3; d++) {
      const float v = p[3 * i + d];
      mn[d] = fminf(mn[d], v);
      mx[d] = fmaxf(mx[d], v);
    }
  float r[6];
  for (int d = 0; d < 6; d++) {
    float v = d < 3 ? mn[d] : -mx[d - 3];
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sh[0];
    for (int w = 1; w < 16; w++) t = fminf(t, sh[w]);
    r[d] = t;
  }
  if (threadIdx.x == 0) {
    GridMeta m;
    float ext = 0.f;
    for (int d = 0; d < 3; d++) ext = fmaxf(ext, (-r[3 + d]) - r[d]);
    float cell = fmaxf(radius, ext / (float)(kGridCap - 1));
    cell = fmaxf(cell, 1e-20f) * 1.0001f;          // strictly larger than the radius: the 3x3x3 block always covers the ball
    m.inv_cell = 1.0f / cell;
    m.ncells = 1;
    for (int d = 0; d < 3; d++) {
      m.org[d] = n > 0 ? r[d] : 0.f;
      int dim = n > 0 ? (int)floorf(((-r[3 + d]) - r[d]) * m.inv_cell) + 1 : 1;
      m.dim[d] = dim < 1 ? 1 : (dim > kGridCap ? kGridCap : dim);
      m.ncells *= m.dim[d];
    }
    G.meta[b] = m;
  }
}

__global__ void grid_count_kernel(const float* __restrict__ s, BatchTable bt, GridLayout G) {
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bt.s_count[b]) return;
  const GridMeta m = G.meta[b];
  const float* p = s + 3 * (bt.s_start[b] + i);
  const int cx = cell_coord(p[0], m.org[0], m.inv_cell, m.dim[0]);
  const int cy = cell_coord(p[1], m.org[1], m.inv_cell, m.dim[1]);
  const int cz = cell_coord(p[2], m.org[2], m.inv_cell, m.dim[2]);
  const int cell = cx + m.dim[0] * (cy + m.dim[1] * cz);
  G.cell_of[bt.s_start[b] + i] = cell;
  atomicAdd(&G.cell_start[(size_t)b * (kCellCap + 1) + cell], 1);
}

__global__ __launch_bounds__(1024) void grid_scan_kernel(GridLayout G) {
  __shared__ int sh[1024];
  const int b = blockIdx.x;
  const int n = G.meta[b].ncells;
  int* a = G.cell_start + (size_t)b * (kCellCap + 1);
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int lo = t * chunk, hi = min(n, lo + chunk);
  int sum = 0;
  for (int i = lo; i < hi; i++) sum += a[i];
  sh[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? sh[t - off] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  int run = sh[t] - sum;
  for (int i = lo; i < hi; i++) {
    const int v = a[i];
    a[i] = run;
    run += v;
  }
  if (t == 1023) a[n] = sh[1023];
}

__global__ void grid_scatter_kernel(const float* __restrict__ s, BatchTable bt, GridLayout G) {
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bt.s_count[b]) return;
  const int64_t g = bt.s_start[b] + i;
  const int cell = G.cell_of[g];
  const int pos = G.cell_start[(size_t)b * (kCellCap + 1) + cell] + atomicAdd(&G.cell_fill[(size_t)b * kCellCap + cell], 1);
  G.sorted[bt.s_start[b] + pos] = make_float4(s[3 * g], s[3 * g + 1], s[3 * g + 2], __int_as_float((int)g));
}

// one wavefront per query; the nine (y, z) rows of the 3x3x3 cell block are contiguous runs of `sorted`
__global__ __launch_bounds__(256) void radius_grid_search_kernel(const float* __restrict__ q, BatchTable bt, GridLayout G,
                                                                 int64_t ns_total, float r2, int limit,
                                                                 int64_t* __restrict__ out, int32_t* __restrict__ max_count,
                                                                 int32_t* __restrict__ tie_rows, int32_t* __restrict__ tie_count) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= bt.q_count[b]) return;
  const GridMeta m = G.meta[b];
  const int64_t gq = bt.q_start[b] + qi;
  const float qx = q[3 * gq], qy = q[3 * gq + 1], qz = q[3 * gq + 2];
  // cell coordinates of the query, NOT clamped: a query outside the support box still has to see the boundary cells
  const int cx = (int)floorf((qx - m.org[0]) * m.inv_cell), cy = (int)floorf((qy - m.org[1]) * m.inv_cell),
            cz = (int)floorf((qz - m.org[2]) * m.inv_cell);
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, m.dim[0] - 1);
  const int* cs = G.cell_start + (size_t)b * (kCellCap + 1);
  const float4* pts = G.sorted + bt.s_start[b];
  unsigned long long best = ~0ull;
  int count = 0;
  // in-radius candidates are compacted into a wave-private LDS buffer; whenever 64 have accumulated they are sorted across the
  // wave (bitonic) and merged into the running 64 smallest -- ~230 instructions per 64 hits instead of ~15 per hit for the
  // one-at-a-time sorted insertion
  __shared__ unsigned long long stage_s[4][128];
  unsigned long long* stage = stage_s[threadIdx.x >> 6];
  int fill = 0;
  // lanes 0..8 fetch the [begin, end) run of one (y, z) row each (all 18 loads in flight at once); the nine runs are then
  // walked as ONE flat candidate list, 64 candidates per step
  int beg = 0, len = 0;
  if (lane < 9 && x0 <= x1) {
    const int z = cz + lane / 3 - 1, y = cy + lane % 3 - 1;
    if (z >= 0 && z < m.dim[2] && y >= 0 && y < m.dim[1]) {
      const int rowc = m.dim[0] * (y + m.dim[1] * z);
      beg = cs[rowc + x0];
      len = cs[rowc + x1 + 1] - beg;
    }
  }
  int rb[9], ro[9], total = 0;
#pragma unroll
  for (int r = 0; r < 9; r++) {
    rb[r] = __builtin_amdgcn_readlane(beg, r);
    ro[r] = total;                                       // exclusive offset of run r in the flat list
    total += __builtin_amdgcn_readlane(len, r);
  }
  for (int base = 0; base < total; base += 64) {
    const int t = base + lane;
    const bool valid = t < total;
    int p = 0;
#pragma unroll
    for (int r = 0; r < 9; r++) p = (t >= ro[r]) ? rb[r] + (t - ro[r]) : p;
    const float4 c = valid ? pts[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float dx = __fsub_rn(qx, c.x), dyv = __fsub_rn(qy, c.y), dzv = __fsub_rn(qz, c.z);
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dyv, dyv)), __fmul_rn(dzv, dzv));
    const bool hit = valid && (d2 < r2);
    const unsigned long long mk = __ballot(hit);
    if (mk == 0ull) continue;
    const int nh = __popcll(mk);
    count += nh;
    if (hit) {
      const int rank = __popcll(mk & ((1ull << lane) - 1ull));
      stage[fill + rank] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(c.w);
    }
    fill += nh;
    if (fill >= 64) {
      const unsigned long long k = wave_sort64(stage[lane]);
      best = wave_merge_lower(best, k);
      const int rem = fill - 64;
      const unsigned long long mv = lane < rem ? stage[64 + lane] : 0ull;
      if (lane < rem) stage[lane] = mv;
      fill = rem;
    }
  }
  if (fill > 0) {
    const unsigned long long k = wave_sort64(lane < fill ? stage[lane] : ~0ull);
    best = wave_merge_lower(best, k);
  }
  if (lane < limit) out[gq * limit + lane] = (best != ~0ull) ? (int64_t)(unsigned)(best & 0xffffffffull) : ns_total;
  // same-line atomics serialise in the L2 (~7 ns each: 80 000 queries = 0.5 ms); the running maximum is monotonic, so a plain
  // (possibly stale) read filters almost all of them
  if (lane == 0 && count > __atomic_load_n(max_count + b, __ATOMIC_RELAXED)) atomicMax(max_count + b, count);
  if (tie_rows != nullptr) flag_tie_row(best, count, limit, gq, tie_rows, tie_count);
}

int fill_batch_table(BatchTable* bt, const int64_t* q_len, const int64_t* s_len, int batch, int64_t nq, int64_t ns,
                     int64_t* qmax, int64_t* smax) {
  int64_t qs = 0, ss = 0;
  *qmax = 0;
  *smax = 0;
  for (int b = 0; b < batch; b++) {
    const int64_t ql = q_len ? q_len[b] : 0, sl = s_len[b];
    if (ql < 0 || sl < 0) return 1;
    bt->q_start[b] = qs; bt->q_count[b] = ql;
    bt->s_start[b] = ss; bt->s_count[b] = sl;
    qs += ql; ss += sl;
    if (ql > *qmax) *qmax = ql;
    if (sl > *smax) *smax = sl;
  }
  return (q_len && qs != nq) || ss != ns;
}

}  // namespace

extern "C" size_t se3_radius_grid_workspace_bytes(int64_t ns, int batch) {
  if (ns < 0 || batch < 1 || batch > SE3_MAX_BATCH) return 0;
  return grid_carve(ns, batch, nullptr, nullptr);
}

extern "C" int se3_radius_grid_build(const float* s_points, int64_t ns, const int64_t* s_lengths_host, int batch, float radius,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(s_points && s_lengths_host && workspace, SE3_ERR_INVALID_ARG, "radius_grid_build: null pointer");
  SE3_REQUIRE(batch >= 1 && batch <= SE3_MAX_BATCH && radius > 0.f, SE3_ERR_INVALID_ARG, "radius_grid_build: bad batch/radius");
  SE3_REQUIRE(ns < (1ll << 31), SE3_ERR_UNSUPPORTED, "radius_grid_build: support too large");
  BatchTable bt;
  int64_t qmax, smax;
  SE3_REQUIRE(fill_batch_table(&bt, nullptr, s_lengths_host, batch, 0, ns, &qmax, &smax) == 0, SE3_ERR_INVALID_ARG,
              "radius_grid_build: lengths do not sum to ns");
  GridLayout G;
  SE3_REQUIRE(grid_carve(ns, batch, (char*)workspace, &G) <= workspace_bytes, SE3_ERR_WORKSPACE,
              "radius_grid_build: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // cell_start and cell_fill are neighbours in the workspace (grid_carve): one fill over both (and the alignment gap between them)
  const size_t span = (size_t)((char*)(G.cell_fill + (size_t)batch * kCellCap) - (char*)G.cell_start);
  if ((char*)G.cell_fill < (char*)G.cell_start || hipMemsetAsync(G.cell_start, 0, span, st) != hipSuccess) {
    se3_set_error("radius_grid_build: memset failed");
    return SE3_ERR_LAUNCH;
  }
  grid_bounds_kernel<<<batch, 1024, 0, st>>>(s_points, bt, radius, G);
  if (smax > 0) {
    dim3 gp((unsigned)se3_cdiv(smax, 256), (unsigned)batch);
    grid_count_kernel<<<gp, 256, 0, st>>>(s_points, bt, G);
    grid_scan_kernel<<<batch, 1024, 0, st>>>(G);
    grid_scatter_kernel<<<gp, 256, 0, st>>>(s_points, bt, G);
  } else {
    grid_scan_kernel<<<batch, 1024, 0, st>>>(G);
  }
  SE3_CHECK_LAUNCH("radius_grid_build");
  return SE3_OK;
}

extern "C" int se3_radius_neighbors_grid_ties(const float* q_points, int64_t nq, const int64_t* q_lengths_host,
                                              const int64_t* s_lengths_host, int64_t ns, int batch, const void* grid_workspace,
                                              float radius, int limit, int64_t* neighbors, int32_t* max_count, int max_count_is_zero,
                                              int32_t* tie_rows, int32_t* tie_count, void* stream) {
  SE3_REQUIRE((tie_rows == nullptr) == (tie_count == nullptr), SE3_ERR_INVALID_ARG, "radius_neighbors_grid: tie_rows and tie_count go together");
  SE3_REQUIRE(q_points && q_lengths_host && s_lengths_host && grid_workspace && neighbors && max_count, SE3_ERR_INVALID_ARG,
              "radius_neighbors_grid: null pointer");
  SE3_REQUIRE(batch >= 1 && batch <= SE3_MAX_BATCH, SE3_ERR_INVALID_ARG, "radius_neighbors_grid: batch");
  SE3_REQUIRE(limit >= 1 && limit <= SE3_MAX_NEIGHBOR_LIMIT, SE3_ERR_UNSUPPORTED, "radius_neighbors_grid: limit %d", limit);
  BatchTable bt;
  int64_t qmax, smax;
  SE3_REQUIRE(fill_batch_table(&bt, q_lengths_host, s_lengths_host, batch, nq, ns, &qmax, &smax) == 0, SE3_ERR_INVALID_ARG,
              "radius_neighbors_grid: lengths do not sum to the sizes");
  GridLayout G;
  grid_carve(ns, batch, (char*)grid_workspace, &G);
  hipStream_t st = (hipStream_t)stream;
  if (!max_count_is_zero && hipMemsetAsync(max_count, 0, sizeof(int32_t) * batch, st) != hipSuccess) {
    se3_set_error("radius_neighbors_grid: memset failed");
    return SE3_ERR_LAUNCH;
  }
  if (nq == 0) return SE3_OK;
  dim3 grid((unsigned)se3_cdiv(qmax, 4), (unsigned)batch);
  radius_grid_search_kernel<<<grid, 256, 0, st>>>(q_points, bt, G, ns, radius * radius, limit, neighbors, max_count, tie_rows, tie_count);
  SE3_CHECK_LAUNCH("radius_neighbors_grid");
  return SE3_OK;
}

extern "C" int se3_radius_neighbors_grid(const float* q_points, int64_t nq, const int64_t* q_lengths_host,
                                         const int64_t* s_lengths_host, int64_t ns, int batch, const void* grid_workspace,
                                         float radius, int limit, int64_t* neighbors, int32_t* max_count, int max_count_is_zero,
                                         void* stream) {
  return se3_radius_neighbors_grid_ties(q_points, nq, q_lengths_host, s_lengths_host, ns, batch, grid_workspace, radius, limit, neighbors,
                                        max_count, max_count_is_zero, nullptr, nullptr, stream);
}

// ---- stacked pairs: the neighbour table cut to the width a batch needs, columns past a PAIR's own width marked -1 ----------------------
// The reference runs one pair per forward and keeps min(limit, that pair's largest count) columns (data.py:96-99 of the reference's
// collate); with several pairs stacked the table is as wide as the widest pair needs and the surplus columns of the narrower pairs are
// marked -1 (every consumer skips them; the padding index Ns would select the zero row instead).  One launch per table instead of a
// column copy and a strided fill per pair.
namespace {
constexpr int kTrimMaxPairs = 64;
struct TrimPairs {
  int n;
  long long row_end[kTrimMaxPairs];
  int width[kTrimMaxPairs];
};
__global__ __launch_bounds__(256) void neighbor_table_trim_kernel(const int64_t* __restrict__ full, int64_t rows, int full_width, int width,
                                                                  TrimPairs P, int64_t* __restrict__ out) {
  const int64_t total = rows * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / width;
    const int col = (int)(i - row * width);
    int w = width;
    for (int p = 0; p < P.n; p++)
      if (row < P.row_end[p]) {
        w = P.width[p];
        break;
      }
    out[i] = col < w ? full[row * full_width + col] : -1;
  }
}
}  // namespace

extern "C" int se3_neighbor_table_trim(const int64_t* full, int64_t rows, int full_width, int width, const int64_t* pair_row_ends_host,
                                       const int* pair_widths_host, int num_pairs, int64_t* out, void* stream) {
  SE3_REQUIRE(full && out && pair_row_ends_host && pair_widths_host, SE3_ERR_INVALID_ARG, "neighbor_table_trim: null pointer");
  SE3_REQUIRE(rows >= 0 && width >= 1 && width <= full_width && num_pairs >= 1 && num_pairs <= kTrimMaxPairs, SE3_ERR_UNSUPPORTED,
              "neighbor_table_trim: rows %lld width %d of %d, %d pairs (max %d)", (long long)rows, width, full_width, num_pairs, kTrimMaxPairs);
  if (rows == 0) return SE3_OK;
  TrimPairs P{};
  P.n = num_pairs;
  for (int p = 0; p < num_pairs; p++) {
    P.row_end[p] = pair_row_ends_host[p];
    P.width[p] = pair_widths_host[p] < width ? pair_widths_host[p] : width;
  }
  neighbor_table_trim_kernel<<<(unsigned)(se3_cdiv(rows * width, 256) < 65536 ? se3_cdiv(rows * width, 256) : 65536), 256, 0, (hipStream_t)stream>>>(full, rows, full_width, width, P, out);
  SE3_CHECK_LAUNCH("neighbor_table_trim");
  return SE3_OK;
}

