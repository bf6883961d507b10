// B2 / D6 row-wise ops: GroupNorm over stacked points (+ residual + LeakyReLU), residual LayerNorm, padded row gather
// and neighbour max pooling.  All are single-pass, HBM-streaming kernels over (rows, C) float32 tensors.
//
//   group norm    geotransformer/modules/e2pn/blocks_epn.py:684-701 (GroupNormEPN: statistics over channels-in-group x
//                 anchors x ALL points of both clouds), kpconv/modules.py:34-51 (GroupNorm on (N, C)); the bottleneck tail
//                 lrelu(norm(x) + shortcut) of blocks_epn.py:838-852 is fused in
//   add+LN        rpe_transformer.py:163-164, vanilla_transformer.py:910-911, output_layer.py:21, 46
//   gather / max  kpconv/functional.py:6-22 (nearest_upsample), e2pn/blocks.py:93-110 (max_pool)
#include "common.h"
#include "group_norm.h"
#include <stdlib.h>

namespace {

constexpr int kGNLanes = 64;   // channels per block
constexpr int kGNRows = 4;     // row lanes per block

__global__ __launch_bounds__(kGNLanes* kGNRows) void gn_partial_kernel(const float* __restrict__ x, SegTable T, int C,
                                                                       float* __restrict__ part) {
  __shared__ WF sh[kGNRows][kGNLanes];
  const int cl = threadIdx.x & (kGNLanes - 1), rl = threadIdx.x / kGNLanes;
  const int c = blockIdx.y * kGNLanes + cl;
  long long r0, r1;
  chunk_rows(T, blockIdx.x, r0, r1);
  WF w = {0.f, 0.f, 0.f};
  if (c < C)
    for (int64_t r = r0 + rl; r < r1; r += kGNRows) {
      const float v = x[r * C + c];
      w.n += 1.f;
      const float d = v - w.mean;
      w.mean += d / w.n;
      w.m2 += d * (v - w.mean);
    }
  sh[rl][cl] = w;
  __syncthreads();
  if (rl == 0 && c < C) {
    for (int k = 1; k < kGNRows; k++) w = wf_merge(w, sh[k][cl]);
    float* p = part + ((int64_t)blockIdx.x * C + c) * 3;
    p[0] = w.n; p[1] = w.mean; p[2] = w.m2;
  }
}

// Throughput version of the partial pass for power-of-two channel counts (16 .. 1024): float4 loads, Q = C / 4 channel quads
// across the block and 256 / Q row lanes, 8 rows in flight per thread, tree merge over the row lanes.  The chunking is a
// pure function of (rows, C), so the result does not depend on scheduling.
// PRE: the statistics are those of lrelu(x scale + shift) -- a GroupNorm + LeakyReLU still pending on x (affine table [segment][2][C]).
// FIN: group partials [chunk][groups][3] (+ bias) instead of channel partials, and the last chunk of a segment to arrive writes the affine
// table itself (group_norm.h: gn_finalize_groups): no finalize launch.
struct GnFinArgs {
  const float* xb;
  const float* gw;
  const float* gb;
  float* affine;      // [segment][2][C]
  int* counters;      // [segment][kGNMaxColumnBlocks], zero between launches
  int groups;
  float eps;
};
template <bool PRE, bool FIN = false>
__global__ __launch_bounds__(256) void gn_partial4_kernel(const float* __restrict__ x, SegTable T, int C,
                                                          float* __restrict__ part, const float* __restrict__ pre_affine = nullptr,
                                                          float pre_slope = 1.f, GnFinArgs fin = GnFinArgs{}) {
  __shared__ WF sh[256][4];
  const int Q = C >> 2, RL = 256 / Q;
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PRE) {
    const float* pa = pre_affine + (size_t)seg_of_chunk(T, blockIdx.x) * 2 * C;
    psc = reinterpret_cast<const float4*>(pa)[q];
    psh = reinterpret_cast<const float4*>(pa + C)[q];
  }
  auto pre = [&](float4 v) {
    if (PRE) {
      v = make_float4(v.x * psc.x + psh.x, v.y * psc.y + psh.y, v.z * psc.z + psh.z, v.w * psc.w + psh.w);
      v.x = v.x > 0.f ? v.x : v.x * pre_slope;
      v.y = v.y > 0.f ? v.y : v.y * pre_slope;
      v.z = v.z > 0.f ? v.z : v.z * pre_slope;
      v.w = v.w > 0.f ? v.w : v.w * pre_slope;
    }
    return v;
  };
  long long r0, r1;
  chunk_rows(T, blockIdx.x, r0, r1);
  WF w[4] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  const float4* xq = reinterpret_cast<const float4*>(x) + q;
  // Two batches of eight rows per row lane in flight (a chunk is 15-20 rows per lane at the sizes of the pyramid: with ONE batch in flight
  // and a remainder loop of single loads the pass was a chain of up to nine memory round trips, 2 TB/s -- tools/micro/gn_stats_rate.py).
  // Rows past the chunk are requested again at the chunk's last row and not counted; the four channels of a lane share their count, so
  // one reciprocal per row serves all four Welford updates (same operations per channel as wf_push: the statistics are bit-identical).
  auto request = [&](float4 (&v)[8], long long r) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const long long row = r + u * RL;
      v[u] = xq[(row < r1 ? row : r1 - 1) * Q];
    }
  };
  auto consume = [&](const float4 (&v)[8], long long r) {
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (r + u * RL < r1) {
        const float4 t = pre(v[u]);
        const float n = w[0].n + 1.f, inv = __frcp_rn(n);
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          w[k].n = n;
          const float d = tv[k] - w[k].mean;
          w[k].mean += d * inv;
          w[k].m2 += d * (tv[k] - w[k].mean);
        }
      }
  };
  {
    float4 va[8], vb[8];
    long long r = r0 + rl;
    if (r < r1) request(va, r);
    while (r < r1) {
      const long long rb = r + 8 * RL, rc = r + 16 * RL;
      if (rb < r1) request(vb, rb);
      consume(va, r);
      if (rb >= r1) break;
      if (rc < r1) request(va, rc);
      consume(vb, rb);
      r = rc;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) sh[threadIdx.x][k] = w[k];
  __syncthreads();
  for (int s = RL >> 1; s > 0; s >>= 1) {
    if (rl < s) {
#pragma unroll
      for (int k = 0; k < 4; k++) sh[threadIdx.x][k] = wf_merge(sh[threadIdx.x][k], sh[threadIdx.x + s * Q][k]);
    }
    __syncthreads();
  }
  if (FIN) {
    WF* shc = &sh[0][0];                                                  // the chunk's channel partials, channel-major (thread q: 4 q .. 4 q + 3)
    if (fin.xb)
      for (int c = threadIdx.x; c < C; c += 256) shc[c].mean += fin.xb[c];
    __syncthreads();
    const int cpg = C / fin.groups;
    for (int g = threadIdx.x; g < fin.groups; g += 256) {
      WF w = shc[g * cpg];
      for (int j = 1; j < cpg; j++) w = wf_merge(w, shc[g * cpg + j]);
      gn_store_partial(part + ((int64_t)blockIdx.x * fin.groups + g) * 3, w);
    }
    const int seg = seg_of_chunk(T, blockIdx.x);
    int cb0 = T.chunk_begin[0], cb1 = T.chunk_begin[1];
#pragma unroll
    for (int i = 1; i < kGNMaxSegments; i++)
      if (seg == i) {
        cb0 = T.chunk_begin[i];
        cb1 = T.chunk_begin[i + 1];
      }
    if (!gn_last_arrival(fin.counters + seg * kGNMaxColumnBlocks, cb1 - cb0)) return;
    gn_finalize_groups(part, fin.groups, cb0, cb1, 0, fin.groups, cpg, fin.xb, fin.gw, fin.gb, fin.eps, fin.affine + (size_t)seg * 2 * C, C);
    return;
  }
  if (rl == 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const WF o = sh[threadIdx.x][k];
      float* p = part + ((int64_t)blockIdx.x * C + 4 * q + k) * 3;
      p[0] = o.n; p[1] = o.mean; p[2] = o.m2;
    }
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ affine_all, SegTable T, int64_t rows, int C,
                                                       int has_slope, float slope, float* __restrict__ y) {
  const int64_t total = rows * C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if ((C & 3) == 0) {
    const int64_t total4 = total >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
      const int64_t row = (i << 2) / C;
      const int c0 = (int)((i << 2) - row * C);
      const float* affine = affine_all + (size_t)(T.n > 1 ? seg_of_row(T, row) : 0) * 2 * C;
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      const float4 sc = *reinterpret_cast<const float4*>(affine + c0), sf = *reinterpret_cast<const float4*>(affine + C + c0);
      float4 o = make_float4(v.x * sc.x + sf.x, v.y * sc.y + sf.y, v.z * sc.z + sf.z, v.w * sc.w + sf.w);
      if (res) {
        const float4 t = reinterpret_cast<const float4*>(res)[i];
        o = make_float4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w);
      }
      if (has_slope) {
        o.x = o.x > 0.f ? o.x : o.x * slope;
        o.y = o.y > 0.f ? o.y : o.y * slope;
        o.z = o.z > 0.f ? o.z : o.z * slope;
        o.w = o.w > 0.f ? o.w : o.w * slope;
      }
      reinterpret_cast<float4*>(y)[i] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int64_t row = i / C;
      const int c = (int)(i - row * C);
      const float* affine = affine_all + (size_t)(T.n > 1 ? seg_of_row(T, row) : 0) * 2 * C;
      float t = x[i] * affine[c] + affine[C + c] + (res ? res[i] : 0.f);
      if (has_slope) t = t > 0.f ? t : t * slope;
      y[i] = t;
    }
  }
}

// The pending forms of the fused dense layers made concrete: y = lrelu_f( Tb(Ta(x)) + R ), T.(v) = lrelu(v scale + shift) per (segment,
// channel), R = residual scale_r + shift_r (a second raw tensor with its own pending GroupNorm: the shortcut of a bottleneck block) or the
// plain residual.  C % 4 == 0.
struct ChainArgs {
  const float* x;
  const float* affine_a;
  const float* affine_b;
  const float* res;
  const float* affine_r;
  float slope_a, slope_b, slope_f;
  int64_t rows;
  int C;
  float* y;
  float* amax;               // blocked / chunked layouts: the largest |y| of the call is atomicMax-ed into this word (or null): the fused KPConv
                             // kernels scale their input by a power of two from it before the f16 split (no magnitude window)
};
// largest magnitude over the workgroup (256 threads) -> at most one atomicMax per workgroup (non-negative floats order like their bit patterns;
// a NaN's pattern is the largest).  Same-line atomics serialise in the L2 (~7 ns each) and one per WAVE of a 16 000-wave launch doubled the
// pass; the running maximum is monotonic, so a plain, possibly stale read filters most of the rest.
__device__ __forceinline__ void amax_commit(float* amax, float m) {
  __shared__ unsigned wave_max[4];
  if (amax == nullptr) return;
  unsigned b = __float_as_uint(m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)b, o);
    b = t > b ? t : b;
  }
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    b = wave_max[0];
    for (int w = 1; w < 4; w++) b = wave_max[w] > b ? wave_max[w] : b;
    if (b > __atomic_load_n(reinterpret_cast<unsigned*>(amax), __ATOMIC_RELAXED)) atomicMax(reinterpret_cast<unsigned*>(amax), b);
  }
}
__device__ __forceinline__ float abs_max4(float m, const float4& o) {
  // (fmaxf drops a NaN beside a number: a NaN / Inf in y is carried as +Inf's pattern or above through the integer maximum instead)
  const unsigned a = __float_as_uint(fabsf(o.x)), b = __float_as_uint(fabsf(o.y)), c = __float_as_uint(fabsf(o.z)), d = __float_as_uint(fabsf(o.w));
  unsigned r = __float_as_uint(m);
  r = a > r ? a : r; r = b > r ? b : r; r = c > r ? c : r; r = d > r ? d : r;
  return __uint_as_float(r);
}
__device__ __forceinline__ float4 affine_lrelu(float4 v, const float* aff, int C, int c0, float slope) {
  const float4 sc = *reinterpret_cast<const float4*>(aff + c0), sf = *reinterpret_cast<const float4*>(aff + C + c0);
  float4 o = make_float4(v.x * sc.x + sf.x, v.y * sc.y + sf.y, v.z * sc.z + sf.z, v.w * sc.w + sf.w);
  o.x = o.x > 0.f ? o.x : o.x * slope;
  o.y = o.y > 0.f ? o.y : o.y * slope;
  o.z = o.z > 0.f ? o.z : o.z * slope;
  o.w = o.w > 0.f ? o.w : o.w * slope;
  return o;
}
__global__ __launch_bounds__(256) void gn_chain_apply_kernel(ChainArgs p, SegTable T) {
  const int64_t total4 = p.rows * p.C >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int C = p.C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int64_t row = (i << 2) / C;
    const int c0 = (int)((i << 2) - row * C);
    const size_t so = (size_t)(T.n > 1 ? seg_of_row(T, row) : 0) * 2 * C;
    float4 o = affine_lrelu(reinterpret_cast<const float4*>(p.x)[i], p.affine_a + so, C, c0, p.slope_a);
    if (p.affine_b) o = affine_lrelu(o, p.affine_b + so, C, c0, p.slope_b);
    if (p.res) {
      float4 t = reinterpret_cast<const float4*>(p.res)[i];
      if (p.affine_r) t = affine_lrelu(t, p.affine_r + so, C, c0, 1.f);
      o = make_float4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w);
    }
    o.x = o.x > 0.f ? o.x : o.x * p.slope_f;
    o.y = o.y > 0.f ? o.y : o.y * p.slope_f;
    o.z = o.z > 0.f ? o.z : o.z * p.slope_f;
    o.w = o.w > 0.f ? o.w : o.w * p.slope_f;
    reinterpret_cast<float4*>(p.y)[i] = o;
  }
}

// The same, written in the layout the fused KPConv gathers from (csrc/kpconv_mfma.hip, BLK): x (points, 6 anchors, C) ->
// [point][C / 16][anchor pair][16 channels][2 anchors].  A thread owns two anchor rows x 4 channels: two 16-byte reads, 32 contiguous bytes out.
__global__ __launch_bounds__(256) void gn_chain_apply_blocked_kernel(ChainArgs p, SegTable T) {
  const int C = p.C, Q = C >> 2;
  const int64_t total = (p.rows >> 1) * Q;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t pr = i / Q;                              // anchor-pair row: point n = pr / 3, pair ap = pr % 3
    const int c0 = (int)(i - pr * Q) << 2;
    const int64_t n = pr / 3;
    const int ap = (int)(pr - n * 3);
    const int64_t r0 = n * 6 + 2 * ap;
    const size_t so = (size_t)(T.n > 1 ? seg_of_row(T, r0) : 0) * 2 * C;
    float4 o[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int64_t at = ((r0 + e) * C + c0) >> 2;
      o[e] = affine_lrelu(reinterpret_cast<const float4*>(p.x)[at], p.affine_a + so, C, c0, p.slope_a);
      if (p.affine_b) o[e] = affine_lrelu(o[e], p.affine_b + so, C, c0, p.slope_b);
      if (p.res) {
        float4 t = reinterpret_cast<const float4*>(p.res)[at];
        if (p.affine_r) t = affine_lrelu(t, p.affine_r + so, C, c0, 1.f);
        o[e] = make_float4(o[e].x + t.x, o[e].y + t.y, o[e].z + t.z, o[e].w + t.w);
      }
      o[e].x = o[e].x > 0.f ? o[e].x : o[e].x * p.slope_f;
      o[e].y = o[e].y > 0.f ? o[e].y : o[e].y * p.slope_f;
      o[e].z = o[e].z > 0.f ? o[e].z : o[e].z * p.slope_f;
      o[e].w = o[e].w > 0.f ? o[e].w : o[e].w * p.slope_f;
    }
    float4* dst = reinterpret_cast<float4*>(p.y + n * 6 * C + (c0 >> 4) * 96 + ap * 32 + (c0 & 15) * 2);
    dst[0] = make_float4(o[0].x, o[1].x, o[0].y, o[1].y);
    dst[1] = make_float4(o[0].z, o[1].z, o[0].w, o[1].w);
    am = abs_max4(abs_max4(am, o[0]), o[1]);
  }
  amax_commit(p.amax, am);
}

// The same, written in the layout the union-staged KPConv loads whole rows from (csrc/kpconv_union.hip, BLK): x (points, 6 anchors, C) ->
// [point][C / 8][6 anchors][8 channels]: the 192 bytes of a point's 8-channel chunk are contiguous.  A thread owns one anchor row x 4 channels,
// indexed by its OUTPUT position (consecutive threads write consecutive 16 bytes; reads are 32-byte pieces).
__global__ __launch_bounds__(256) void gn_chain_apply_chunked_kernel(ChainArgs p, SegTable T) {
  const int C = p.C, Q = 6 * C >> 2;                       // float4 per point
  const int64_t total = (p.rows / 6) * Q;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t n = i / Q;
    const int rem = (int)(i - n * Q);
    const int chunk = rem / 12, q = rem - chunk * 12;
    const int a = q >> 1, c0 = chunk * 8 + (q & 1) * 4;
    const int64_t r = n * 6 + a;
    const size_t so = (size_t)(T.n > 1 ? seg_of_row(T, r) : 0) * 2 * C;
    const int64_t at = (r * C + c0) >> 2;
    float4 o = affine_lrelu(reinterpret_cast<const float4*>(p.x)[at], p.affine_a + so, C, c0, p.slope_a);
    if (p.affine_b) o = affine_lrelu(o, p.affine_b + so, C, c0, p.slope_b);
    if (p.res) {
      float4 t = reinterpret_cast<const float4*>(p.res)[at];
      if (p.affine_r) t = affine_lrelu(t, p.affine_r + so, C, c0, 1.f);
      o = make_float4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w);
    }
    o.x = o.x > 0.f ? o.x : o.x * p.slope_f;
    o.y = o.y > 0.f ? o.y : o.y * p.slope_f;
    o.z = o.z > 0.f ? o.z : o.z * p.slope_f;
    o.w = o.w > 0.f ? o.w : o.w * p.slope_f;
    reinterpret_cast<float4*>(p.y)[i] = o;
    am = abs_max4(am, o);
  }
  amax_commit(p.amax, am);
}

// ---- GroupNorm backward (training step) ------------------------------------------------------------------------------------------------
// y = lrelu(xhat w + b [+ res]),  xhat = (x + xb - mean_g) rstd_g  per (segment, group).  With dz = dy lrelu'(.):
//   dres = dz,   db[c] = sum_r dz,   dw[c] = sum_r dz xhat,
//   dx   = rstd (w dz - mean_g(w dz) - xhat mean_g(w dz xhat)) = a[c] dz + e x + f[c],   dxb[c] = sum_r dx.
// Three passes: the forward statistics again (partial + finalize), per-chunk channel sums (sum dz, sum dz x, sum x), apply.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ dout, const float* __restrict__ affine_all,
                                                             SegTable T, int C, int has_slope, float slope, float* __restrict__ part) {
  __shared__ float sh[4][64][3];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  long long r0, r1;
  chunk_rows(T, blockIdx.x, r0, r1);
  const float* affine = affine_all + (size_t)seg_of_chunk(T, blockIdx.x) * 2 * C;
  float s1 = 0.f, t2 = 0.f, s4 = 0.f;
  if (c < C) {
    const float sc = affine[c], sf = affine[C + c];
#pragma unroll 8
    for (long long r = r0 + rl; r < r1; r += 4) {
      const float xv = x[r * C + c];
      float dz = dout[r * C + c];
      if (has_slope) {
        const float pre = xv * sc + sf + (res ? res[r * C + c] : 0.f);
        dz = pre > 0.f ? dz : dz * slope;
      }
      s1 += dz;
      t2 = fmaf(dz, xv, t2);
      s4 += xv;
    }
  }
  sh[rl][cl][0] = s1; sh[rl][cl][1] = t2; sh[rl][cl][2] = s4;
  __syncthreads();
  if (rl == 0 && c < C) {
    float* p = part + ((int64_t)blockIdx.x * C + c) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = (sh[0][cl][k] + sh[1][cl][k]) + (sh[2][cl][k] + sh[3][cl][k]);
  }
}

// per (group, segment): channel sums over the segment's chunks -> the coefficients of the apply pass and the segment's contribution to
// (dweight, dbias, dxbias): params[seg][0..2][C]
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ xb,
                                                              const float* __restrict__ gw, const float* __restrict__ stats, int C,
                                                              int groups, SegTable T, float* __restrict__ coef,
                                                              float* __restrict__ params) {
  __shared__ float red[2][256];
  const int g = blockIdx.x, cpg = C / groups, seg = blockIdx.y;
  int cb = T.chunk_begin[0], ce = T.chunk_begin[1];
  long long rb = T.row_begin[0], re = T.row_begin[1];
#pragma unroll
  for (int i = 1; i < kGNMaxSegments; i++)
    if (seg == i) {
      cb = T.chunk_begin[i]; ce = T.chunk_begin[i + 1];
      rb = T.row_begin[i]; re = T.row_begin[i + 1];
    }
  const float* st = stats + ((size_t)seg * groups + g) * 3;
  const float mean = st[0], rstd = st[1], n = st[2], nrows = (float)(re - rb);
  coef += (size_t)seg * 3 * C;
  params += (size_t)seg * 3 * C;
  float a1 = 0.f, a2 = 0.f;                     // this thread's share of sum_c w S1, sum_c w S2
  // channel sums over the segment's chunks: with few channels per group the 256 threads split the chunks (fixed order: deterministic)
  __shared__ float csum[256][3];
  const bool split = cpg <= 256 && 256 % cpg == 0;
  if (split) {
    const int ci = threadIdx.x % cpg, lanes = 256 / cpg, cl = threadIdx.x / cpg;
    float s1 = 0.f, t2 = 0.f, s4 = 0.f;
    for (int ch = cb + cl; ch < ce; ch += lanes) {
      const float* p = part + ((int64_t)ch * C + g * cpg + ci) * 3;
      s1 += p[0]; t2 += p[1]; s4 += p[2];
    }
    csum[threadIdx.x][0] = s1; csum[threadIdx.x][1] = t2; csum[threadIdx.x][2] = s4;
    __syncthreads();
    for (int st = lanes >> 1; st > 0; st >>= 1) {          // lanes is a power of two whenever cpg divides 256
      if (cl < st) {
#pragma unroll
        for (int k = 0; k < 3; k++) csum[threadIdx.x][k] += csum[threadIdx.x + st * cpg][k];
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < cpg; i += 256) {
    const int c = g * cpg + i;
    float s1 = 0.f, t2 = 0.f, s4 = 0.f;
    if (split) {
      s1 = csum[i][0]; t2 = csum[i][1]; s4 = csum[i][2];
    } else {
      for (int ch = cb; ch < ce; ch++) {
        const float* p = part + ((int64_t)ch * C + c) * 3;
        s1 += p[0]; t2 += p[1]; s4 += p[2];
      }
    }
    const float shift = (xb ? xb[c] : 0.f) - mean;
    const float s2 = rstd * (t2 + shift * s1);                 // sum dz xhat
    params[c] = s2;                                            // dweight
    params[C + c] = s1;                                        // dbias
    params[2 * C + c] = s4;                                    // sum x: turned into dxbias below
    a1 = fmaf(gw[c], s1, a1);
    a2 = fmaf(gw[c], s2, a2);
  }
  red[0][threadIdx.x] = a1; red[1][threadIdx.x] = a2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  const float m1 = red[0][0] / n, m2 = red[1][0] / n;
  for (int i = threadIdx.x; i < cpg; i += 256) {
    const int c = g * cpg + i;
    const float shift = (xb ? xb[c] : 0.f) - mean;
    const float a = rstd * gw[c], e = -rstd * rstd * m2, f = -rstd * m1 + e * shift;
    coef[c] = a; coef[C + c] = e; coef[2 * C + c] = f;
    params[2 * C + c] = a * params[C + c] + e * params[2 * C + c] + f * nrows;     // dxbias = sum_r dx
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ dout, const float* __restrict__ affine_all,
                                                           const float* __restrict__ coef_all, SegTable T, int64_t rows, int C,
                                                           int has_slope, float slope, float* __restrict__ dx,
                                                           float* __restrict__ dres) {
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / C;
    const int c = (int)(i - row * C);
    const int seg = T.n > 1 ? seg_of_row(T, row) : 0;
    const float* affine = affine_all + (size_t)seg * 2 * C;
    const float* coef = coef_all + (size_t)seg * 3 * C;
    const float xv = x[i];
    float dz = dout[i];
    if (has_slope) {
      const float pre = xv * affine[c] + affine[C + c] + (res ? res[i] : 0.f);
      dz = pre > 0.f ? dz : dz * slope;
    }
    dx[i] = coef[c] * dz + coef[C + c] * xv + coef[2 * C + c];
    if (dres) dres[i] = dz;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm(hidden + residual): one wavefront per row; residual row index = row % res_rows (broadcast over anchors)
// ---------------------------------------------------------------------------------------------------------------------
template <int VPL>   // float4 vectors per lane: C <= 256 * VPL
__global__ __launch_bounds__(256) void add_ln_kernel(const float* __restrict__ h, const float* __restrict__ hb,
                                                     const float* __restrict__ res, const float* __restrict__ w,
                                                     const float* __restrict__ b, int64_t rows, int64_t res_rows, int C,
                                                     float eps, float* __restrict__ y) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int C4 = C >> 2;
  const float4* hp = reinterpret_cast<const float4*>(h + row * C);
  const float4* rp = reinterpret_cast<const float4*>(res + (row % res_rows) * C);
  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; k++) {
    const int i = lane + 64 * k;
    if (i < C4) {
      const float4 a = hp[i], r = rp[i];
      const float4 c = hb ? reinterpret_cast<const float4*>(hb)[i] : make_float4(0.f, 0.f, 0.f, 0.f);   // bias of the producing linear
      v[k] = make_float4((a.x + c.x) + r.x, (a.y + c.y) + r.y, (a.z + c.z) + r.z, (a.w + c.w) + r.w);
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    } else {
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = se3_wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; k++) {
    const int i = lane + 64 * k;
    if (i < C4) {
      const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rstd = 1.0f / sqrtf(se3_wave_sum(q) / (float)C + eps);
  float4* yp = reinterpret_cast<float4*>(y + row * C);
#pragma unroll
  for (int k = 0; k < VPL; k++) {
    const int i = lane + 64 * k;
    if (i < C4) {
      const float4 ww = reinterpret_cast<const float4*>(w)[i], bb = reinterpret_cast<const float4*>(b)[i];
      yp[i] = make_float4((v[k].x - mean) * rstd * ww.x + bb.x, (v[k].y - mean) * rstd * ww.y + bb.y,
                          (v[k].z - mean) * rstd * ww.z + bb.z, (v[k].w - mean) * rstd * ww.w + bb.w);
    }
  }
}

// Backward of add_ln_kernel: v = h + hb + res, xhat = (v - mean) rstd, y = xhat w + b.  dv = rstd (dy w - mean(dy w) - xhat mean(dy w xhat));
// dh = dv (also the residual's gradient; summed over the anchor blocks by the caller when the residual was broadcast); per-channel sums
// params[0] = sum_rows dy xhat (d weight), params[1] = sum_rows dy (d bias), params[2] = sum_rows dv (d hidden_bias): a wave walks 8 rows with the
// sums in registers, the four waves of a workgroup meet in LDS, one float atomic per channel, sum and workgroup (params zero-initialised).
template <int VPL, bool PARTIALS = false>      // PARTIALS: params is [workgroup][3][C], written, not accumulated (se3_add_layer_norm_bwd_partials)
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const float* __restrict__ h, const float* __restrict__ hb,
                                                         const float* __restrict__ res, const float* __restrict__ w,
                                                         const float* __restrict__ dy, int64_t rows, int64_t res_rows, int C, float eps,
                                                         float* __restrict__ dh, float* __restrict__ params) {
  __shared__ float4 red[3][4][64 * VPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  float4 sw[VPL], sb[VPL], sv[VPL];
#pragma unroll
  for (int k = 0; k < VPL; k++) sw[k] = sb[k] = sv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int rr = 0; rr < 8; rr++) {
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 8 + rr;
    if (row >= rows) break;                                  // wave-uniform
    const float4* hp = reinterpret_cast<const float4*>(h + row * C);
    const float4* rp = reinterpret_cast<const float4*>(res + (row % res_rows) * C);
    const float4* gp = reinterpret_cast<const float4*>(dy + row * C);
    float4 v[VPL], g[VPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; k++) {
      const int i = lane + 64 * k;
      v[k] = g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < C4) {
        const float4 a = hp[i], r = rp[i];
        const float4 c = hb ? reinterpret_cast<const float4*>(hb)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        v[k] = make_float4((a.x + c.x) + r.x, (a.y + c.y) + r.y, (a.z + c.z) + r.z, (a.w + c.w) + r.w);
        g[k] = gp[i];
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      }
    }
    const float mean = se3_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; k++) {
      const int i = lane + 64 * k;
      if (i < C4) {
        v[k] = make_float4(v[k].x - mean, v[k].y - mean, v[k].z - mean, v[k].w - mean);
        q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
      }
    }
    const float rstd = 1.0f / sqrtf(se3_wave_sum(q) / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;                               // sum dy w, sum dy w xhat
    float4 gw[VPL];
#pragma unroll
    for (int k = 0; k < VPL; k++) {
      const int i = lane + 64 * k;
      gw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < C4) {
        const float4 ww = reinterpret_cast<const float4*>(w)[i];
        v[k] = make_float4(v[k].x * rstd, v[k].y * rstd, v[k].z * rstd, v[k].w * rstd);      // xhat
        gw[k] = make_float4(g[k].x * ww.x, g[k].y * ww.y, g[k].z * ww.z, g[k].w * ww.w);
        m1 += (gw[k].x + gw[k].y) + (gw[k].z + gw[k].w);
        m2 += (gw[k].x * v[k].x + gw[k].y * v[k].y) + (gw[k].z * v[k].z + gw[k].w * v[k].w);
      }
    }
    m1 = se3_wave_sum(m1) / (float)C;
    m2 = se3_wave_sum(m2) / (float)C;
    float4* dp = reinterpret_cast<float4*>(dh + row * C);
#pragma unroll
    for (int k = 0; k < VPL; k++) {
      const int i = lane + 64 * k;
      if (i < C4) {
        const float4 d = make_float4(rstd * (gw[k].x - m1 - v[k].x * m2), rstd * (gw[k].y - m1 - v[k].y * m2),
                                     rstd * (gw[k].z - m1 - v[k].z * m2), rstd * (gw[k].w - m1 - v[k].w * m2));
        dp[i] = d;
        sw[k] = make_float4(sw[k].x + g[k].x * v[k].x, sw[k].y + g[k].y * v[k].y, sw[k].z + g[k].z * v[k].z, sw[k].w + g[k].w * v[k].w);
        sb[k] = make_float4(sb[k].x + g[k].x, sb[k].y + g[k].y, sb[k].z + g[k].z, sb[k].w + g[k].w);
        sv[k] = make_float4(sv[k].x + d.x, sv[k].y + d.y, sv[k].z + d.z, sv[k].w + d.w);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < VPL; k++) {
    red[0][wave][lane + 64 * k] = sw[k];
    red[1][wave][lane + 64 * k] = sb[k];
    red[2][wave][lane + 64 * k] = sv[k];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * C; e += 256) {
    const int which = e / C, c = e - which * C;
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; wv++) t += reinterpret_cast<const float*>(&red[which][wv][0])[c];
    if (PARTIALS) params[(size_t)blockIdx.x * 3 * C + e] = t;                  // per-block sums: the caller adds them in block order (deterministic)
    else unsafeAtomicAdd(params + e, t);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// out[i, :] = x[idx[i], :] (zeros when idx[i] == n)   and   out[i, :] = max_j xpad[idx[i, j], :]
// ---------------------------------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int64_t n, int64_t m,
                                   int64_t width, float* __restrict__ out) {
  const int64_t total = m * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / width, c = i - r * width;
    const int64_t s = idx[r];
    out[i] = (s >= 0 && s < n) ? x[s * width + c] : 0.f;
  }
}

__global__ __launch_bounds__(256) void neighbor_max_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                           int64_t n, int64_t m, int nn, int64_t width, float* __restrict__ out) {
  // The REAL neighbours of the point, compacted (the tables of the fine stages are mostly padding: 34 % / 55 % / 86 % real entries in the
  // three pooling tables of the 5k-point pyramid, tools/micro/neighbor_fill.py): element offsets of their rows, the list filled up to a
  // multiple of 8 with its last row (a repeated row does not change a maximum).  A padded entry (index n: the reference's zero row)
  // contributes 0 once; a column beyond the pair's own table width (-1, several pairs stacked) contributes nothing.
  __shared__ unsigned row_s[64];
  __shared__ int cnt_s, pad_s;
  const int64_t r = blockIdx.x;
  if (threadIdx.x < 64) {
    const int j = threadIdx.x;
    const int64_t s = j < nn ? idx[r * nn + j] : -1;
    const bool real = s >= 0 && s < n;
    const unsigned long long mask = __ballot(real), pads = __ballot(s >= n);
    const int cnt = __popcll(mask);
    if (real) row_s[__popcll(mask & ((1ull << j) - 1ull))] = (unsigned)(s * width);
    const int last = 63 - __clzll((long long)(mask | 1ull));              // lane of the last real entry (lane 0 when there is none)
    const unsigned last_row = (unsigned)__shfl(real ? (unsigned)(s * width) : 0u, last);
    const int cnt8 = (cnt + 7) & ~7;
    if (j >= cnt && j < cnt8) row_s[j] = last_row;
    if (j == 0) {
      cnt_s = cnt;
      pad_s = pads != 0ull;
    }
  }
  __syncthreads();
  const int cnt = cnt_s, cnt8 = (cnt + 7) & ~7;
  const float floor_v = pad_s ? 0.f : -INFINITY;
  // 8 gathered rows in flight per thread (the loop was one L2 round trip per neighbour)
  if ((width & 3) == 0) {
    const int64_t w4 = width >> 2;
    for (int64_t c = threadIdx.x; c < w4; c += blockDim.x) {
      float4 best = make_float4(floor_v, floor_v, floor_v, floor_v);
      for (int j0 = 0; j0 < cnt8; j0 += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = reinterpret_cast<const float4*>(x + row_s[j0 + u])[c];
#pragma unroll
        for (int u = 0; u < 8; u++)
          best = make_float4(fmaxf(best.x, v[u].x), fmaxf(best.y, v[u].y), fmaxf(best.z, v[u].z), fmaxf(best.w, v[u].w));
      }
      reinterpret_cast<float4*>(out + r * width)[c] = best;
    }
  } else {
    for (int64_t c = threadIdx.x; c < width; c += blockDim.x) {
      float best = floor_v;
      for (int j0 = 0; j0 < cnt8; j0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = x[row_s[j0 + u] + c];
#pragma unroll
        for (int u = 0; u < 8; u++) best = fmaxf(best, v[u]);
      }
      out[r * width + c] = best;
    }
  }
}

// Backward of neighbor_max_kernel: the gradient of out[r, c] goes to the neighbour that holds the maximum (the first one in table order on
// ties, as torch.max(dim) picks a single index; nothing when the maximum is the zero row of a padded entry).  dx is accumulated with
// hardware float atomics (several rows share a neighbour) and must be zero-initialised by the caller.
template <bool FIXED = false>      // FIXED: 64-bit fixed-point sums (common.h: order-independent, bit-identical runs), dxf zero on entry
__global__ __launch_bounds__(256) void neighbor_max_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                               const float* __restrict__ dout, int64_t n, int64_t m, int nn,
                                                               int64_t width, float* __restrict__ dx, const float* __restrict__ bound = nullptr,
                                                               unsigned long long* __restrict__ dxf = nullptr) {
  const double fscale = FIXED ? ldexp(1.0, se3_fixed_scale_exp(bound, m, 0)) : 1.0;
  __shared__ int64_t nb[64];
  const int64_t r = blockIdx.x;
  for (int j = threadIdx.x; j < nn; j += blockDim.x) nb[j] = idx[r * nn + j];
  __syncthreads();
  for (int64_t c = threadIdx.x; c < width; c += blockDim.x) {
    float best = -INFINITY;
    int64_t arg = -1;
    for (int j = 0; j < nn; j++) {
      const int64_t s = nb[j];
      if (s < 0) continue;
      const float v = s < n ? x[s * width + c] : 0.f;
      if (v > best) {
        best = v;
        arg = s;
      }
    }
    if (arg >= 0 && arg < n) {
      if constexpr (FIXED) se3_fixed_add(dxf + arg * width + c, dout[r * width + c], fscale);
      else unsafeAtomicAdd(dx + arg * width + c, dout[r * width + c]);
    }
  }
}

// out[r, c] = max_a x[a * anchor_stride + r * row_stride + c]: the maximum over the anchor axis of (A, R, C) or (R, A, C) features
// (InvOutBlockEPN, blocks_epn.py:908-926; the 'amax' between equivariant and invariant transformer blocks,
// conditional_transformer.py:282-283,299-302).  float4 per thread, the A reads of a thread in flight together.
template <int A>
__global__ __launch_bounds__(256) void anchor_max_kernel(const float* __restrict__ x, int64_t rows, int C4, int64_t anchor_stride,
                                                         int64_t row_stride, float* __restrict__ out) {
  const int64_t total = rows * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C4;
    const int c4 = (int)(i - r * C4);
    const float* p = x + r * row_stride + 4 * c4;
    float4 v[A];
#pragma unroll
    for (int a = 0; a < A; a++) v[a] = *reinterpret_cast<const float4*>(p + a * anchor_stride);
    float4 m = v[0];
#pragma unroll
    for (int a = 1; a < A; a++) m = make_float4(fmaxf(m.x, v[a].x), fmaxf(m.y, v[a].y), fmaxf(m.z, v[a].z), fmaxf(m.w, v[a].w));
    reinterpret_cast<float4*>(out)[i] = m;
  }
}

inline unsigned grid_for(int64_t work, int tpb) {
  int64_t g = se3_cdiv(work, tpb);
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" size_t se3_group_norm_workspace_bytes(int64_t rows, int channels, int groups) {
  (void)rows;
  (void)groups;   // upper bound over any segmentation: <= kGNMaxChunks + 16 chunk partials, 16 affine tables
  return (size_t)((kGNMaxChunks + kGNMaxSegments) * channels * 3 + kGNMaxSegments * 2 * channels) * sizeof(float) + 256;
}

extern "C" int se3_group_norm_segments_fwd(const float* x, const float* x_bias, const float* residual, const float* weight,
                                           const float* bias, int64_t rows, int channels, int groups,
                                           const int64_t* segment_row_offsets_host, int num_segments, float eps,
                                           int apply_leaky_relu, float slope, float* out, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(x && weight && bias && out && workspace, SE3_ERR_INVALID_ARG, "group_norm: null pointer");
  SE3_REQUIRE(rows >= 1 && channels >= 1 && groups >= 1 && channels % groups == 0, SE3_ERR_INVALID_ARG,
              "group_norm: rows %lld channels %d groups %d", (long long)rows, channels, groups);
  SE3_REQUIRE(num_segments >= 1 && num_segments <= kGNMaxSegments && (num_segments == 1 || segment_row_offsets_host),
              SE3_ERR_UNSUPPORTED, "group_norm: %d segments (1..%d)", num_segments, kGNMaxSegments);
  SE3_REQUIRE(workspace_bytes >= se3_group_norm_workspace_bytes(rows, channels, groups), SE3_ERR_WORKSPACE,
              "group_norm: workspace too small");
  SegTable T{};
  T.n = num_segments;
  // up to 4 blocks per CU in total for large tensors (a single block per CU is latency bound: 1 TB/s at 100 MB)
  const int cap = gn_chunk_cap(num_segments);
  int chunks = 0;
  for (int sgm = 0; sgm < num_segments; sgm++) {
    const int64_t b0 = num_segments == 1 ? 0 : segment_row_offsets_host[sgm];
    const int64_t b1 = num_segments == 1 ? rows : segment_row_offsets_host[sgm + 1];
    SE3_REQUIRE(b1 > b0 && b0 >= 0 && b1 <= rows, SE3_ERR_INVALID_ARG, "group_norm: segment %d rows [%lld, %lld)", sgm,
                (long long)b0, (long long)b1);
    T.row_begin[sgm] = b0;
    T.row_begin[sgm + 1] = b1;
    T.chunk_begin[sgm] = chunks;
    chunks += (int)gn_chunks(b1 - b0, channels, cap);
    T.chunk_begin[sgm + 1] = chunks;
  }
  SE3_REQUIRE(T.row_begin[0] == 0 && T.row_begin[num_segments] == rows, SE3_ERR_INVALID_ARG,
              "group_norm: the segments must cover all rows");
  float* part = (float*)workspace;
  float* affine = part + (size_t)(kGNMaxChunks + kGNMaxSegments) * channels * 3;
  hipStream_t st = (hipStream_t)stream;
  if (gn_fast_path(channels) && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    gn_partial4_kernel<false><<<(unsigned)chunks, 256, 0, st>>>(x, T, channels, part);
  } else {
    dim3 g1((unsigned)chunks, (unsigned)se3_cdiv(channels, kGNLanes));
    gn_partial_kernel<<<g1, kGNLanes * kGNRows, 0, st>>>(x, T, channels, part);
  }
  gn_finalize_kernel<<<dim3((unsigned)groups, (unsigned)num_segments), 256, 0, st>>>(part, x_bias, weight, bias, channels,
                                                                                      groups, T, eps, affine);
  const int64_t work = (channels % 4 == 0) ? rows * channels / 4 : rows * channels;
  gn_apply_kernel<<<grid_for(work, 256), 256, 0, st>>>(x, residual, affine, T, rows, channels, apply_leaky_relu, slope, out);
  SE3_CHECK_LAUNCH("group_norm");
  return SE3_OK;
}

// The two halves of se3_group_norm_segments_fwd on their own, for activations that stay in their pending form (dense_norm.hip):
// statistics -> affine table [segment][2][channels] (scale, shift) without a pass that applies it ...
static int gn_segment_table(const char* who, int64_t rows, int channels, const int64_t* segment_row_offsets_host, int num_segments, bool chunked,
                            SegTable& T, int& chunks) {
  SE3_REQUIRE(num_segments >= 1 && num_segments <= kGNMaxSegments && (num_segments == 1 || segment_row_offsets_host), SE3_ERR_UNSUPPORTED,
              "%s: %d segments (1..%d)", who, num_segments, kGNMaxSegments);
  T = SegTable{};
  T.n = num_segments;
  const int cap = gn_chunk_cap(num_segments);
  chunks = 0;
  for (int sgm = 0; sgm < num_segments; sgm++) {
    const int64_t b0 = num_segments == 1 ? 0 : segment_row_offsets_host[sgm];
    const int64_t b1 = num_segments == 1 ? rows : segment_row_offsets_host[sgm + 1];
    SE3_REQUIRE(b1 > b0 && b0 >= 0 && b1 <= rows, SE3_ERR_INVALID_ARG, "%s: segment %d rows [%lld, %lld)", who, sgm, (long long)b0, (long long)b1);
    T.row_begin[sgm] = b0;
    T.row_begin[sgm + 1] = b1;
    T.chunk_begin[sgm] = chunks;
    if (chunked) chunks += (int)gn_chunks(b1 - b0, channels, cap);
    T.chunk_begin[sgm + 1] = chunks;
  }
  SE3_REQUIRE(T.row_begin[0] == 0 && T.row_begin[num_segments] == rows, SE3_ERR_INVALID_ARG, "%s: the segments must cover all rows", who);
  return SE3_OK;
}

extern "C" size_t se3_group_norm_stats_workspace_bytes(int channels) {
  return kGNCounterB + se3_group_norm_workspace_bytes(0, channels, 1);
}

extern "C" int se3_group_norm_stats(const float* x, const float* in_affine, float in_slope, const float* x_bias, const float* weight,
                                    const float* bias, int64_t rows, int channels, int groups, const int64_t* segment_row_offsets_host,
                                    int num_segments, float eps, float* affine_out, void* workspace, size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(x && weight && bias && affine_out && workspace, SE3_ERR_INVALID_ARG, "group_norm_stats: null pointer");
  SE3_REQUIRE(rows >= 1 && channels >= 1 && groups >= 1 && channels % groups == 0, SE3_ERR_INVALID_ARG,
              "group_norm_stats: rows %lld channels %d groups %d", (long long)rows, channels, groups);
  SE3_REQUIRE(workspace_bytes >= se3_group_norm_stats_workspace_bytes(channels), SE3_ERR_WORKSPACE, "group_norm_stats: workspace too small");
  const bool fast = gn_fast_path(channels) && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  SE3_REQUIRE(fast || !in_affine, SE3_ERR_UNSUPPORTED, "group_norm_stats: a pending input form needs a power-of-two channel count (16..1024), got %d",
              channels);
  SegTable T;
  int chunks;
  const int rc = gn_segment_table("group_norm_stats", rows, channels, segment_row_offsets_host, num_segments, true, T, chunks);
  if (rc != SE3_OK) return rc;
  // [arrival counters (zero before the first call, left zero by every call)][partials]
  float* part = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + kGNCounterB);
  hipStream_t st = (hipStream_t)stream;
  if (fast) {
    const GnFinArgs fin{x_bias, weight, bias, affine_out, static_cast<int*>(workspace), groups, eps};
    if (in_affine) gn_partial4_kernel<true, true><<<(unsigned)chunks, 256, 0, st>>>(x, T, channels, part, in_affine, in_slope, fin);
    else gn_partial4_kernel<false, true><<<(unsigned)chunks, 256, 0, st>>>(x, T, channels, part, nullptr, 1.f, fin);
  } else {
    gn_partial_kernel<<<dim3((unsigned)chunks, (unsigned)se3_cdiv(channels, kGNLanes)), kGNLanes * kGNRows, 0, st>>>(x, T, channels, part);
    gn_finalize_kernel<<<dim3((unsigned)groups, (unsigned)num_segments), 256, 0, st>>>(part, x_bias, weight, bias, channels, groups, T, eps, affine_out);
  }
  SE3_CHECK_LAUNCH("group_norm_stats");
  return SE3_OK;
}

// ... and the pass that makes a pending form concrete: out = lrelu_f( Tb(Ta(x)) + R ) (gn_chain_apply_kernel); slopes of 1 = no LeakyReLU.
extern "C" int se3_group_norm_apply(const float* x, const float* affine_a, float slope_a, const float* affine_b, float slope_b,
                                    const float* residual, const float* residual_affine, float final_slope, int64_t rows, int channels,
                                    const int64_t* segment_row_offsets_host, int num_segments, int blocked_layout, float* out, void* stream) {
  return se3_group_norm_apply_amax(x, affine_a, slope_a, affine_b, slope_b, residual, residual_affine, final_slope, rows, channels,
                                   segment_row_offsets_host, num_segments, blocked_layout, out, nullptr, stream);
}

extern "C" int se3_group_norm_apply_amax(const float* x, const float* affine_a, float slope_a, const float* affine_b, float slope_b,
                                         const float* residual, const float* residual_affine, float final_slope, int64_t rows, int channels,
                                         const int64_t* segment_row_offsets_host, int num_segments, int blocked_layout, float* out,
                                         float* amax_out, void* stream) {
  SE3_REQUIRE(x && affine_a && out, SE3_ERR_INVALID_ARG, "group_norm_apply: null pointer");
  SE3_REQUIRE(amax_out == nullptr || blocked_layout != 0, SE3_ERR_UNSUPPORTED, "group_norm_apply: amax_out is written by the blocked / chunked layouts only");
  SE3_REQUIRE(rows >= 1 && channels >= 4 && channels % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, SE3_ERR_UNSUPPORTED,
              "group_norm_apply: rows %lld channels %d (a multiple of 4)", (long long)rows, channels);
  SE3_REQUIRE(residual || !residual_affine, SE3_ERR_INVALID_ARG, "group_norm_apply: residual_affine without a residual");
  SegTable T;
  int chunks;
  const int rc = gn_segment_table("group_norm_apply", rows, channels, segment_row_offsets_host, num_segments, false, T, chunks);
  if (rc != SE3_OK) return rc;
  SE3_REQUIRE(blocked_layout != 1 || (channels % 16 == 0 && rows % 6 == 0), SE3_ERR_UNSUPPORTED,
              "group_norm_apply: the blocked layout is for (points, 6, channels) with channels %% 16 == 0 (rows %lld, channels %d)", (long long)rows,
              channels);
  SE3_REQUIRE(blocked_layout != 2 || (channels % 8 == 0 && rows % 6 == 0), SE3_ERR_UNSUPPORTED,
              "group_norm_apply: the chunked layout is for (points, 6, channels) with channels %% 8 == 0 (rows %lld, channels %d)", (long long)rows,
              channels);
  SE3_REQUIRE(blocked_layout >= 0 && blocked_layout <= 2, SE3_ERR_INVALID_ARG, "group_norm_apply: blocked_layout %d", blocked_layout);
  ChainArgs p{x, affine_a, affine_b, residual, residual_affine, slope_a, slope_b, final_slope, rows, channels, out, amax_out};
  // (with the magnitude word: at most 1024 workgroups, i.e. at most 1024 -- mostly filtered -- atomics; the loops are grid-strided)
  const unsigned cap = amax_out ? 1024u : 4096u;
  if (blocked_layout == 2) gn_chain_apply_chunked_kernel<<<min(grid_for(rows * channels / 4, 256), cap), 256, 0, (hipStream_t)stream>>>(p, T);
  else if (blocked_layout) gn_chain_apply_blocked_kernel<<<min(grid_for(rows * channels / 8, 256), cap), 256, 0, (hipStream_t)stream>>>(p, T);
  else gn_chain_apply_kernel<<<grid_for(rows * channels / 4, 256), 256, 0, (hipStream_t)stream>>>(p, T);
  SE3_CHECK_LAUNCH("group_norm_apply");
  return SE3_OK;
}

// Backward of se3_group_norm_segments_fwd.  grad_x (rows, C); grad_residual (rows, C) or NULL; grad_params (num_segments, 3, C): every
// segment's contribution to (d weight, d bias, d x_bias) -- the caller sums over segments.  Workspace: se3_group_norm_bwd_workspace_bytes.
extern "C" size_t se3_group_norm_bwd_workspace_bytes(int channels) {
  return se3_group_norm_workspace_bytes(0, channels, 1) + (size_t)((kGNMaxChunks + kGNMaxSegments) * channels * 3 + kGNMaxSegments * 6 * channels) * sizeof(float);
}

extern "C" int se3_group_norm_segments_bwd(const float* x, const float* x_bias, const float* residual, const float* weight,
                                           const float* bias, const float* grad_out, int64_t rows, int channels, int groups,
                                           const int64_t* segment_row_offsets_host, int num_segments, float eps, int apply_leaky_relu,
                                           float slope, float* grad_x, float* grad_residual, float* grad_params, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(x && weight && bias && grad_out && grad_x && grad_params && workspace, SE3_ERR_INVALID_ARG, "group_norm_bwd: null pointer");
  SE3_REQUIRE(rows >= 1 && channels >= 1 && groups >= 1 && channels % groups == 0, SE3_ERR_INVALID_ARG,
              "group_norm_bwd: rows %lld channels %d groups %d", (long long)rows, channels, groups);
  SE3_REQUIRE(num_segments >= 1 && num_segments <= kGNMaxSegments && (num_segments == 1 || segment_row_offsets_host),
              SE3_ERR_UNSUPPORTED, "group_norm_bwd: %d segments (1..%d)", num_segments, kGNMaxSegments);
  SE3_REQUIRE(workspace_bytes >= se3_group_norm_bwd_workspace_bytes(channels), SE3_ERR_WORKSPACE, "group_norm_bwd: workspace too small");
  SegTable T{};
  T.n = num_segments;
  const int cap = gn_chunk_cap(num_segments);
  int chunks = 0;
  for (int sgm = 0; sgm < num_segments; sgm++) {
    const int64_t b0 = num_segments == 1 ? 0 : segment_row_offsets_host[sgm];
    const int64_t b1 = num_segments == 1 ? rows : segment_row_offsets_host[sgm + 1];
    SE3_REQUIRE(b1 > b0 && b0 >= 0 && b1 <= rows, SE3_ERR_INVALID_ARG, "group_norm_bwd: segment %d rows [%lld, %lld)", sgm,
                (long long)b0, (long long)b1);
    T.row_begin[sgm] = b0;
    T.row_begin[sgm + 1] = b1;
    T.chunk_begin[sgm] = chunks;
    chunks += (int)gn_chunks(b1 - b0, channels, cap);
    T.chunk_begin[sgm + 1] = chunks;
  }
  SE3_REQUIRE(T.row_begin[0] == 0 && T.row_begin[num_segments] == rows, SE3_ERR_INVALID_ARG,
              "group_norm_bwd: the segments must cover all rows");
  const size_t part_floats = (size_t)(kGNMaxChunks + kGNMaxSegments) * channels * 3;
  float* part = (float*)workspace;
  float* affine = part + part_floats;
  float* stats = affine + (size_t)kGNMaxSegments * 2 * channels;       // (segments, groups, 3): groups <= channels
  float* bpart = stats + (size_t)kGNMaxSegments * 3 * channels + 64;
  float* coef = bpart + part_floats;
  hipStream_t st = (hipStream_t)stream;
  // 1. the forward statistics
  if (gn_fast_path(channels) && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    gn_partial4_kernel<false><<<(unsigned)chunks, 256, 0, st>>>(x, T, channels, part);
  } else {
    dim3 g1((unsigned)chunks, (unsigned)se3_cdiv(channels, kGNLanes));
    gn_partial_kernel<<<g1, kGNLanes * kGNRows, 0, st>>>(x, T, channels, part);
  }
  gn_finalize_kernel<<<dim3((unsigned)groups, (unsigned)num_segments), 256, 0, st>>>(part, x_bias, weight, bias, channels, groups, T, eps,
                                                                                      affine, stats);
  // 2. channel sums of the incoming gradient, 3. coefficients, 4. apply
  gn_bwd_partial_kernel<<<dim3((unsigned)chunks, (unsigned)se3_cdiv(channels, 64)), 256, 0, st>>>(x, residual, grad_out, affine, T, channels,
                                                                                                 apply_leaky_relu, slope, bpart);
  gn_bwd_finalize_kernel<<<dim3((unsigned)groups, (unsigned)num_segments), 256, 0, st>>>(bpart, x_bias, weight, stats, channels, groups, T,
                                                                                          coef, grad_params);
  gn_bwd_apply_kernel<<<grid_for(rows * channels, 256), 256, 0, st>>>(x, residual, grad_out, affine, coef, T, rows, channels,
                                                                     apply_leaky_relu, slope, grad_x, grad_residual);
  SE3_CHECK_LAUNCH("group_norm_bwd");
  return SE3_OK;
}

extern "C" int se3_group_norm_fwd(const float* x, const float* x_bias, const float* residual, const float* weight,
                                  const float* bias, int64_t rows, int channels, int groups, float eps, int apply_leaky_relu,
                                  float slope, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return se3_group_norm_segments_fwd(x, x_bias, residual, weight, bias, rows, channels, groups, nullptr, 1, eps,
                                     apply_leaky_relu, slope, out, workspace, workspace_bytes, stream);
}

extern "C" int se3_add_layer_norm_fwd(const float* hidden, const float* hidden_bias, const float* residual,
                                      const float* weight, const float* bias, int64_t rows, int64_t residual_rows,
                                      int channels, float eps, float* out, void* stream) {
  SE3_REQUIRE(hidden && residual && weight && bias && out, SE3_ERR_INVALID_ARG, "add_layer_norm: null pointer");
  SE3_REQUIRE(channels % 4 == 0 && channels >= 4 && channels <= 2048, SE3_ERR_UNSUPPORTED,
              "add_layer_norm: channels %d (need a multiple of 4 up to 2048)", channels);
  SE3_REQUIRE(rows >= 0 && residual_rows >= 1 && rows % residual_rows == 0, SE3_ERR_INVALID_ARG,
              "add_layer_norm: rows %lld not a multiple of residual rows %lld", (long long)rows, (long long)residual_rows);
  if (rows == 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)se3_cdiv(rows, 4);
  if (channels <= 256)
    add_ln_kernel<1><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, bias, rows, residual_rows, channels, eps, out);
  else if (channels <= 512)
    add_ln_kernel<2><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, bias, rows, residual_rows, channels, eps, out);
  else if (channels <= 1024)
    add_ln_kernel<4><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, bias, rows, residual_rows, channels, eps, out);
  else
    add_ln_kernel<8><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, bias, rows, residual_rows, channels, eps, out);
  SE3_CHECK_LAUNCH("add_layer_norm");
  return SE3_OK;
}

// The same with per-workgroup partial sums instead of float atomics: grad_params_partials is (se3_add_layer_norm_bwd_blocks(rows), 3, channels),
// every element written; the caller adds the blocks in order (a deterministic reduction: two runs are bit-identical).
extern "C" int64_t se3_add_layer_norm_bwd_blocks(int64_t rows) { return se3_cdiv(rows, 32); }
extern "C" int se3_add_layer_norm_bwd_partials(const float* hidden, const float* hidden_bias, const float* residual, const float* weight,
                                               const float* grad_out, int64_t rows, int64_t residual_rows, int channels, float eps,
                                               float* grad_hidden, float* grad_params_partials, void* stream) {
  SE3_REQUIRE(hidden && residual && weight && grad_out && grad_hidden && grad_params_partials, SE3_ERR_INVALID_ARG, "add_layer_norm_bwd: null pointer");
  SE3_REQUIRE(channels >= 4 && channels % 4 == 0 && channels <= 2048 && residual_rows >= 1 && rows % residual_rows == 0, SE3_ERR_UNSUPPORTED,
              "add_layer_norm_bwd: channels %d (multiple of 4, <= 2048), rows %lld / residual rows %lld", channels, (long long)rows,
              (long long)residual_rows);
  if (rows == 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)se3_cdiv(rows, 32);
  if (channels <= 256)
    add_ln_bwd_kernel<1, true><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, grad_out, rows, residual_rows, channels, eps, grad_hidden, grad_params_partials);
  else if (channels <= 512)
    add_ln_bwd_kernel<2, true><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, grad_out, rows, residual_rows, channels, eps, grad_hidden, grad_params_partials);
  else if (channels <= 1024)
    add_ln_bwd_kernel<4, true><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, grad_out, rows, residual_rows, channels, eps, grad_hidden, grad_params_partials);
  else
    add_ln_bwd_kernel<8, true><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, grad_out, rows, residual_rows, channels, eps, grad_hidden, grad_params_partials);
  SE3_CHECK_LAUNCH("add_layer_norm_bwd");
  return SE3_OK;
}

// Backward of se3_add_layer_norm_fwd: grad_hidden (rows, channels) (= the gradient of the un-broadcast residual), grad_params (3, channels) =
// (d weight, d bias, d hidden_bias), zero-initialised by the caller (float atomics).
extern "C" int se3_add_layer_norm_bwd(const float* hidden, const float* hidden_bias, const float* residual, const float* weight,
                                      const float* grad_out, int64_t rows, int64_t residual_rows, int channels, float eps,
                                      float* grad_hidden, float* grad_params, void* stream) {
  SE3_REQUIRE(hidden && residual && weight && grad_out && grad_hidden && grad_params, SE3_ERR_INVALID_ARG, "add_layer_norm_bwd: null pointer");
  SE3_REQUIRE(channels >= 4 && channels % 4 == 0 && channels <= 2048 && residual_rows >= 1 && rows % residual_rows == 0, SE3_ERR_UNSUPPORTED,
              "add_layer_norm_bwd: channels %d (multiple of 4, <= 2048), rows %lld / residual rows %lld", channels, (long long)rows,
              (long long)residual_rows);
  if (rows == 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)se3_cdiv(rows, 32);
  if (channels <= 256)
    add_ln_bwd_kernel<1><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, grad_out, rows, residual_rows, channels, eps, grad_hidden, grad_params);
  else if (channels <= 512)
    add_ln_bwd_kernel<2><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, grad_out, rows, residual_rows, channels, eps, grad_hidden, grad_params);
  else if (channels <= 1024)
    add_ln_bwd_kernel<4><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, grad_out, rows, residual_rows, channels, eps, grad_hidden, grad_params);
  else
    add_ln_bwd_kernel<8><<<grid, 256, 0, st>>>(hidden, hidden_bias, residual, weight, grad_out, rows, residual_rows, channels, eps, grad_hidden, grad_params);
  SE3_CHECK_LAUNCH("add_layer_norm_bwd");
  return SE3_OK;
}

extern "C" int se3_gather_rows_padded(const float* x, const int64_t* idx, int64_t n, int64_t m, int64_t width, float* out,
                                      void* stream) {
  SE3_REQUIRE(x && idx && out, SE3_ERR_INVALID_ARG, "gather_rows_padded: null pointer");
  if (m * width == 0) return SE3_OK;
  gather_rows_kernel<<<grid_for(m * width, 256), 256, 0, (hipStream_t)stream>>>(x, idx, n, m, width, out);
  SE3_CHECK_LAUNCH("gather_rows_padded");
  return SE3_OK;
}

extern "C" int se3_anchor_max(const float* x, int num_anchors, int64_t rows, int channels, int64_t anchor_stride,
                              int64_t row_stride, float* out, void* stream) {
  SE3_REQUIRE(x && out, SE3_ERR_INVALID_ARG, "anchor_max: null pointer");
  SE3_REQUIRE(num_anchors == 6 && channels >= 4 && channels % 4 == 0 && anchor_stride % 4 == 0 && row_stride % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
              SE3_ERR_UNSUPPORTED, "anchor_max: %d anchors (6), %d channels (multiple of 4), 16-byte aligned rows", num_anchors,
              channels);
  if (rows <= 0) return SE3_OK;
  const int64_t work = rows * (channels / 4);
  const int64_t blocks = se3_cdiv(work, 256);
  anchor_max_kernel<6><<<(unsigned)(blocks > 16384 ? 16384 : blocks), 256, 0, (hipStream_t)stream>>>(x, rows, channels / 4,
                                                                                                  anchor_stride, row_stride, out);
  SE3_CHECK_LAUNCH("anchor_max");
  return SE3_OK;
}

extern "C" int se3_neighbor_max_pool_bwd(const float* x, const int64_t* idx, const float* dout, int64_t n, int64_t m, int nn,
                                         int64_t width, float* dx, void* stream) {
  SE3_REQUIRE(x && idx && dout && dx && nn >= 1 && nn <= 64, SE3_ERR_INVALID_ARG, "neighbor_max_pool_bwd: bad arguments (nn <= 64)");
  if (m * width == 0) return SE3_OK;
  const int threads = width >= 256 ? 256 : (width >= 128 ? 128 : 64);
  neighbor_max_bwd_kernel<false><<<(unsigned)m, threads, 0, (hipStream_t)stream>>>(x, idx, dout, n, m, nn, width, dx);
  SE3_CHECK_LAUNCH("neighbor_max_pool_bwd");
  return SE3_OK;
}

// The same with order-independent 64-bit fixed-point sums: dx_fixed (n, width) int64, ZERO on entry; max_abs_dout: DEVICE word (any upper bound
// of |dout|); se3_fixed_to_float(dx_fixed, n * width, max_abs_dout, m, 0, dx) converts.
extern "C" int se3_neighbor_max_pool_bwd_fixed(const float* x, const int64_t* idx, const float* dout, int64_t n, int64_t m, int nn,
                                               int64_t width, const float* max_abs_dout, long long* dx_fixed, void* stream) {
  SE3_REQUIRE(x && idx && dout && max_abs_dout && dx_fixed && nn >= 1 && nn <= 64, SE3_ERR_INVALID_ARG, "neighbor_max_pool_bwd_fixed: bad arguments (nn <= 64)");
  if (m * width == 0) return SE3_OK;
  const int threads = width >= 256 ? 256 : (width >= 128 ? 128 : 64);
  neighbor_max_bwd_kernel<true><<<(unsigned)m, threads, 0, (hipStream_t)stream>>>(x, idx, dout, n, m, nn, width, nullptr, max_abs_dout,
                                                                                  reinterpret_cast<unsigned long long*>(dx_fixed));
  SE3_CHECK_LAUNCH("neighbor_max_pool_bwd_fixed");
  return SE3_OK;
}

// Transpose of se3_gather_rows_padded (its backward): dx_fixed[idx[i], :] += g[i, :] for idx[i] in [0, n) as 64-bit fixed-point sums (order-
// independent); dx_fixed (n, width) int64 ZERO on entry; se3_fixed_to_float(dx_fixed, n * width, max_abs_g, m, 0, dx) converts.
namespace {
__global__ __launch_bounds__(256) void scatter_add_rows_fixed_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx, int64_t n, int64_t m,
                                                                     int64_t width, const float* __restrict__ bound,
                                                                     unsigned long long* __restrict__ dxf) {
  const double fscale = ldexp(1.0, se3_fixed_scale_exp(bound, m, 0));
  const int64_t total = m * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / width, c = i - r * width, j = idx[r];
    if (j >= 0 && j < n) se3_fixed_add(dxf + j * width + c, g[i], fscale);
  }
}
}  // namespace
extern "C" int se3_scatter_add_rows_fixed(const float* g, const int64_t* idx, int64_t n, int64_t m, int64_t width, const float* max_abs_g,
                                          long long* dx_fixed, void* stream) {
  SE3_REQUIRE(g && idx && max_abs_g && dx_fixed, SE3_ERR_INVALID_ARG, "scatter_add_rows_fixed: null pointer");
  if (m * width == 0) return SE3_OK;
  scatter_add_rows_fixed_kernel<<<grid_for(m * width, 256), 256, 0, (hipStream_t)stream>>>(g, idx, n, m, width, max_abs_g,
                                                                                          reinterpret_cast<unsigned long long*>(dx_fixed));
  SE3_CHECK_LAUNCH("scatter_add_rows_fixed");
  return SE3_OK;
}

extern "C" int se3_neighbor_max_pool(const float* x, const int64_t* idx, int64_t n, int64_t m, int nn, int64_t width,
                                     float* out, void* stream) {
  SE3_REQUIRE(x && idx && out && nn >= 1 && nn <= 64, SE3_ERR_INVALID_ARG, "neighbor_max_pool: bad arguments (nn <= 64)");
  if (m * width == 0) return SE3_OK;
  const int threads = width >= 1024 ? 256 : (width >= 256 ? 128 : 64);
  neighbor_max_kernel<<<(unsigned)m, threads, 0, (hipStream_t)stream>>>(x, idx, n, m, nn, width, out);
  SE3_CHECK_LAUNCH("neighbor_max_pool");
  return SE3_OK;
}


// ---- eq2inv_soft for all pairs of a batch (conditional_transformer.py:209-249): out[a, r, :] = sum_e mix[pair(r)][a, e] x[e, r, :] ----------
// feats1_inv[a] = sum_r w0[r] feats1[trace[r, a]] with the 24 rotation weights collapsed onto the (A, A) matrix `mix` of the pair the packed
// row r belongs to (rows outside every pair: zero).  One launch instead of an einsum + a copy per pair.
namespace {
struct MixPairs {
  int n;
  int start[16], length[16];
};
__global__ __launch_bounds__(256) void anchor_mix_stack_kernel(const float* __restrict__ x, int64_t R, int C4, const float* __restrict__ mix,
                                                               MixPairs P, float* __restrict__ out) {
  const int64_t total = R * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C4;
    int p = -1;
    for (int q = 0; q < P.n; q++)
      if (r >= P.start[q] && r < P.start[q] + P.length[q]) p = q;
    float4 v[6];
#pragma unroll
    for (int e = 0; e < 6; e++) v[e] = *reinterpret_cast<const float4*>(x + (e * R * C4 + i) * 4);
#pragma unroll
    for (int a = 0; a < 6; a++) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p >= 0) {
#pragma unroll
        for (int e = 0; e < 6; e++) {
          const float w = mix[(p * 6 + a) * 6 + e];
          o.x += w * v[e].x; o.y += w * v[e].y; o.z += w * v[e].z; o.w += w * v[e].w;
        }
      }
      *reinterpret_cast<float4*>(out + (a * R * C4 + i) * 4) = o;
    }
  }
}
}  // namespace

extern "C" int se3_anchor_mix_stack(const float* x, int64_t rows, int channels, const float* mix, const int64_t* starts, const int64_t* lengths,
                                    int num_pairs, float* out, void* stream) {
  SE3_REQUIRE(x && mix && starts && lengths && out, SE3_ERR_INVALID_ARG, "anchor_mix_stack: null pointer");
  SE3_REQUIRE(num_pairs >= 1 && num_pairs <= 16 && channels % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(out) & 15) == 0,
              SE3_ERR_UNSUPPORTED, "anchor_mix_stack: %d pairs (1..16), %d channels (a multiple of 4)", num_pairs, channels);
  if (rows == 0) return SE3_OK;
  MixPairs P{};
  P.n = num_pairs;
  for (int p = 0; p < num_pairs; p++) {
    P.start[p] = (int)starts[p];
    P.length[p] = (int)lengths[p];
  }
  const int64_t work = rows * (channels / 4);
  anchor_mix_stack_kernel<<<(unsigned)(se3_cdiv(work, 256) > 8192 ? 8192 : se3_cdiv(work, 256)), 256, 0, (hipStream_t)stream>>>(
      x, rows, channels / 4, mix, P, out);
  SE3_CHECK_LAUNCH("anchor_mix_stack");
  return SE3_OK;
}
