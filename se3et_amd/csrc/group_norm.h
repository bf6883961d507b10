// GroupNorm over stacked points: the pieces shared by the row-wise kernels (rowops.hip) and the dense layer that produces GroupNorm
// statistics in its epilogue (dense_norm.hip).  geotransformer/modules/e2pn/blocks_epn.py:684-701 (GroupNormEPN).
#pragma once
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm: per-(row chunk, channel) Welford partials -> per-group finalize (Chan merge) -> normalize.
// Deterministic (no atomics) and cancellation-free (M2 form).
// ---------------------------------------------------------------------------------------------------------------------
struct WF {
  float n, mean, m2;
};
__device__ __forceinline__ WF wf_merge(WF a, WF b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  WF r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean;
  const float f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}

// Segments: independent row ranges of one stacked tensor, each with its own statistics (one registration pair each when
// several pairs share a launch; the reference normalises per pair because it runs one pair per forward).
constexpr int kGNMaxSegments = 16;
constexpr int kGNMaxChunks = 1024;
struct SegTable {
  int n;
  int quantum;                             // rows per chunk are a multiple of this (0 / 1: any); the dense layer's chunks hold whole row tiles
  long long row_begin[kGNMaxSegments + 1];
  int chunk_begin[kGNMaxSegments + 1];
};
__device__ __forceinline__ int seg_of_chunk(const SegTable& T, int chunk) {
  int s = 0;
#pragma unroll
  for (int i = 1; i < kGNMaxSegments; i++)
    if (i < T.n && chunk >= T.chunk_begin[i]) s = i;
  return s;
}
__device__ __forceinline__ int seg_of_row(const SegTable& T, long long row) {
  int s = 0;
#pragma unroll
  for (int i = 1; i < kGNMaxSegments; i++)
    if (i < T.n && row >= T.row_begin[i]) s = i;
  return s;
}
// the rows [r0, r1) of global chunk `chunk`
__device__ __forceinline__ void chunk_rows(const SegTable& T, int chunk, long long& r0, long long& r1) {
  const int s = seg_of_chunk(T, chunk);
  long long b0 = T.row_begin[0], b1 = T.row_begin[1];
  int c0 = T.chunk_begin[0], c1 = T.chunk_begin[1];
#pragma unroll
  for (int i = 1; i < kGNMaxSegments; i++)
    if (s == i) {
      b0 = T.row_begin[i];
      b1 = T.row_begin[i + 1];
      c0 = T.chunk_begin[i];
      c1 = T.chunk_begin[i + 1];
    }
  long long per = (b1 - b0 + (c1 - c0) - 1) / (c1 - c0);
  if (T.quantum > 1) per = (per + T.quantum - 1) / T.quantum * T.quantum;
  r0 = b0 + (long long)(chunk - c0) * per;
  r1 = min(b1, r0 + per);
}

__device__ __forceinline__ void wf_push(WF& w, float v) {
  w.n += 1.f;
  const float d = v - w.mean;
  w.mean += d * __frcp_rn(w.n);
  w.m2 += d * (v - w.mean);
}


// xb (may be null): per-channel constant added to x before the normalisation (the bias of the linear layer that produced x);
// it shifts the per-channel partial means and leaves the M2 terms unchanged, so only this pass sees it.  Output: the affine
// map of every channel, y = x * scale[c] + shift[c] with scale = rstd_g w[c], shift = b[c] + (xb[c] - mean_g) scale.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ xb,
                                                          const float* __restrict__ gw, const float* __restrict__ gb, int C,
                                                          int groups, SegTable T, float eps, float* __restrict__ affine,
                                                          float* __restrict__ stats = nullptr) {
  __shared__ WF sh[256];
  __shared__ float mean_s, rstd_s;
  const int g = blockIdx.x, cpg = C / groups, seg = blockIdx.y;
  int cb = T.chunk_begin[0], ce = T.chunk_begin[1];
#pragma unroll
  for (int i = 1; i < kGNMaxSegments; i++)
    if (seg == i) {
      cb = T.chunk_begin[i];
      ce = T.chunk_begin[i + 1];
    }
  affine += (size_t)seg * 2 * C;
  const int total = (ce - cb) * cpg;
  WF w = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < total; i += 256) {
    const int chunk = cb + i / cpg, c = g * cpg + (i % cpg);
    const float* p = part + ((int64_t)chunk * C + c) * 3;
    WF o = {p[0], p[1] + (xb ? xb[c] : 0.f), p[2]};
    w = wf_merge(w, o);
  }
  sh[threadIdx.x] = w;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] = wf_merge(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const WF r = sh[0];
    mean_s = r.mean;
    rstd_s = 1.0f / sqrtf(r.m2 / r.n + eps);
    if (stats) {                               // (mean of x + xb, rstd, element count) of (segment, group): the backward pass reads them
      float* st = stats + ((size_t)seg * groups + g) * 3;
      st[0] = mean_s; st[1] = rstd_s; st[2] = r.n;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cpg; i += 256) {
    const int c = g * cpg + i;
    const float scale = rstd_s * gw[c];
    affine[c] = scale;
    affine[C + c] = gb[c] + ((xb ? xb[c] : 0.f) - mean_s) * scale;
  }
}

// In-kernel finalize of the producers that deliver GROUP partials [chunk][groups][3] (the bias already inside the means): the workgroup
// that finds itself last of its segment (arrival counter) calls this with all of its 4 waves.  One wave per group: lanes merge the
// segment's chunks [cb0, cb1), a butterfly merges the lanes, lanes < channels-per-group write the affine map of their channel.
constexpr int kGNMaxColumnBlocks = 16;
constexpr size_t kGNCounterB = kGNMaxSegments * kGNMaxColumnBlocks * sizeof(int);
__device__ __forceinline__ void gn_store_partial(float* p, const WF& w) {          // read back by another compute unit in the same launch
  __hip_atomic_store(p + 0, w.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p + 1, w.mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p + 2, w.m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Memory model note: the hand-off below is the sc1 / sc1 form of /opt/skills/guides/MI355X_MICROARCH.md, "Valid forms
// besides Guideline 16's R1/R2" -- every store of the handed-off partials is an agent-scope (sc1, write-through) store, every storing wave
// drains them with `s_waitcnt vmcnt(0)`, the ONE signalling lane adds to the counter behind the workgroup barrier that follows those waits,
// and every load of the partials is an agent-scope (sc1) load issued only by the workgroup whose add returned the last ticket, behind a
// workgroup barrier -- the row "ONE lane of each storing workgroup ... an agent-scope atomic add / the workgroup whose add came last" of that
// guide's table (measured valid on gfx950 / ROCm 7.2; not an architectural guarantee: tools/micro/buffer_sc1_soffset.hip pins the sc1
// lowering this relies on).  A release / acquire fence pair would be the portable form and costs a write-back + invalidate of the XCD's L2
// per workgroup (1.7 -> 3.3 ms over the unary shapes).  The counters must be zero before a launch: the host clears the
// workspace when a call returns an error (se3et_amd/ops.py::_check_counters).
// true in every thread of the workgroup that arrives last of `expected`; resets the counter for the next launch.  No agent-scope fence:
// on this chip that is a write-back and an invalidate of the whole L2 of the XCD (every workgroup of a streaming kernel doing one halves
// the kernel's rate); the partials are agent-scope atomic stores and loads (they go through to the memory side on their own), so all that
// is needed is that this workgroup's stores have completed before its ticket is drawn.
__device__ __forceinline__ bool gn_last_arrival(int* counter, int expected) {
  __shared__ int ticket;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (a workgroup-scope release fence compiles to nothing here)
  __syncthreads();
  if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != expected - 1) return false;
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
// All 256 threads: T = 256 / (groups rounded up to a power of two) threads per group (at most 64, at least 1), each merging every T-th
// chunk of the segment with its loads issued 16 at a time (the partials were written by other compute units: every load is a trip to the
// memory side, and a chain of dependent ones is what made a one-wave-per-group version cost 25 us on a one-pair tensor), then a
// butterfly over the T lanes.
__device__ __forceinline__ WF gn_load_partial(const float* p) {
  return WF{__hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
            __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
}
__device__ __forceinline__ void gn_finalize_groups(const float* part, int groups, int cb0, int cb1, int g_begin, int g_count, int cpg,
                                                   const float* xb, const float* gw, const float* gb, float eps, float* affine, int C) {
  int gp2 = 1;
  while (gp2 < g_count) gp2 <<= 1;
  const int T = gp2 >= 256 ? 1 : (256 / gp2 > 64 ? 64 : 256 / gp2);        // threads per group (a power of two: lanes of one wave)
  const int per_pass = 256 / T;                                            // groups handled at once
  const int sub = threadIdx.x % T;
  for (int gi = threadIdx.x / T; gi < gp2; gi += per_pass) {
    const bool live = gi < g_count;
    const int g = g_begin + (live ? gi : 0);
    WF w = {0.f, 0.f, 0.f};
    for (int i0 = cb0 + sub; i0 < cb1; i0 += 16 * T) {
      WF o[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int i = i0 + j * T;
        o[j] = (live && i < cb1) ? gn_load_partial(part + ((int64_t)i * groups + g) * 3) : WF{0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 16; j++) w = wf_merge(w, o[j]);
    }
    for (int d = 1; d < T; d <<= 1) {
      const WF o = {__shfl_xor(w.n, d), __shfl_xor(w.mean, d), __shfl_xor(w.m2, d)};
      w = wf_merge(w, o);
    }
    if (!live) continue;
    // the lanes of a group merged in different orders (round-off): lane 0 of the group decides, the others take its values
    const float mean = w.mean, rstd = 1.0f / sqrtf(w.m2 / fmaxf(w.n, 1.f) + eps);
    const float mean0 = __shfl(mean, (threadIdx.x & 63) - sub), rstd0 = __shfl(rstd, (threadIdx.x & 63) - sub);
    for (int j = sub; j < cpg; j += T) {
      const int c = g * cpg + j;
      const float scale = rstd0 * gw[c];
      affine[c] = scale;
      affine[C + c] = gb[c] + ((xb ? xb[c] : 0.f) - mean0) * scale;
    }
  }
}

}  // namespace

static bool gn_fast_path(int channels) { return channels >= 16 && channels <= 1024 && (channels & (channels - 1)) == 0; }

// chunks per segment: 256 in all (one workgroup per compute unit), shared by the segments.  More chunks only move time into the merge of
// the last workgroup to arrive, a chain of memory round trips behind the whole pass: with 1 024 chunks over the 8 segments of a batch the
// statistics passes of a step took 0.59 ms, with 512 0.45, with 256 0.43 (tools/micro/gn_stats_rate.py; a one-pair tensor: 11 us with
// 1 024 chunks, 6 with 256).
// The merge works per SEGMENT, so a tensor of one segment takes at most 128 (a one-pair tensor of 30 000 rows: 18.3 us with 256 chunks, 14.0
// with 128; 128 chunks in all for the 8-segment batch are too few workgroups for the stream: 0.54 ms).
constexpr int kGNStatChunks = 256, kGNSegmentChunks = 128;
static int gn_chunk_cap(int num_segments) {
  int cap = kGNStatChunks / num_segments;
  if (cap > kGNSegmentChunks) cap = kGNSegmentChunks;
  return cap > 8 ? cap : 8;
}

// row chunks of the partial pass: a pure function of the problem size (the statistics must not depend on scheduling)
static int64_t gn_chunks(int64_t rows, int channels, int cap) {
  int64_t n;
  if (gn_fast_path(channels)) {
    const int64_t row_lanes = 256 / (channels / 4);
    n = rows / (8 * row_lanes) + 1;           // >= 8 rows per row lane = one batch of loads in flight (32 rows: 20-30 % slower on the
                                              // one-pair-per-forward tensors, where the partial pass is one latency chain per thread)
  } else {
    n = rows / 64 + 1;
  }
  return n > cap ? cap : n;
}

