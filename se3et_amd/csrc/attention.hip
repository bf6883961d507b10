// D1-D5: attention kernels of the SE3ET coarse transformer on gfx950 (fp32 in, fp32 accumulate, f32 MFMA).
//
//   rpe_bias_kernel    the HBM-streaming half of RPE self attention (rpe_transformer.py:39-131): for every query row n the
//                      relative-position term of the logits, bias[ah, n, m] = qp[n, ah, :] . E[n, m, :] (+ qe . Eeq), where
//                      qp = W_p^T q is the position projection folded onto the query side.  The (N, M, C) embedding is read
//                      exactly once, in 16-row tiles of fully used 64 B segments, straight into the B operand of
//                      v_mfma_f32_16x16x4_f32; the 24 (anchor, head) folded queries of the row sit in LDS in MFMA-fragment
//                      order (ds_read_b128, conflict free).  Never materialises p = W_p E (N, M, C) or eq = W_eq Eeq.
//   attention_kernel   softmax((q k^T [+ bias]) * scale) v per (anchor, head, 32-query tile), flash style: S^T = K Q^T with
//                      v_mfma_f32_32x32x2_f32 so that a lane owns ONE query column (row max / row sum are lane-local plus one
//                      cross-half exchange), and the P^T accumulator registers are directly the B operand of O^T += V^T P^T.
//                      The 4 waves of a workgroup split the key tiles and merge (m, l, O) through LDS.  Serves RPE self
//                      attention (with bias), plain cross attention (vanilla_transformer.py:39-85, optionally per-anchor
//                      values) and, looped over key anchors, the equivariant cross attention.
//   cross_eq_stats     g[a, e] = sum_{n,m} (mean_h q_a.k_e / sqrt(d))^2 (vanilla_transformer.py:380-389,425-426), partial
//                      sums per workgroup (deterministic two-stage reduction).
//   cross_eq_apply     out[a] = sum_e W[a, e] softmax_m(S[a, e]) v_e (vanilla_transformer.py:812-818; r_soft collapsed from
//                      24 rotations to the (A, A) anchor pairs, :506-577,839-845).
#include <hip/hip_ext.h>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float f4get(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// ---------------------------------------------------------------------------------------------------------------------
// stack mode: the clouds of a pair (ref, src) share one launch
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kMaxClouds = SE3_MAX_BATCH;        // 32 clouds = 16 pairs per launch (descriptors: 48 B each in the kernel-argument segment)
struct StackCloud {
  const float* emb;     // (N, M, C) geometric embedding (bias kernel only)
  const float* eq;      // (A, N, M, 4) equivariant embedding or null
  int q_start, k_start; // first packed row of the cloud's queries / keys
  int N, M, Mp;         // queries, keys, logits row stride (multiple of 32 >= M)
  int unit_begin;       // bias kernel: flat index of the cloud's first (row, 32-key unit)
  long long bias_off;   // float offset of the cloud's (A*H, N, Mp) logits block
};
struct Stack {
  StackCloud c[kMaxClouds];
  int n;
  int total_units;
};

// idx is wave-uniform: the descriptor is read from the kernel-argument segment with scalar loads at a computed offset
__device__ __forceinline__ StackCloud stack_pick(const Stack& S, int idx) {
  const StackCloud& c = S.c[idx];
  StackCloud r;
  r.emb = c.emb;
  r.eq = c.eq;
  r.q_start = c.q_start;
  r.k_start = c.k_start;
  r.N = c.N;
  r.M = c.M;
  r.Mp = c.Mp;
  r.unit_begin = c.unit_begin;
  r.bias_off = c.bias_off;
  return r;
}

// =====================================================================================================================
// rpe_bias_kernel
// =====================================================================================================================
// round-to-nearest-even bf16 of x, and of the remainder x - hi (finite inputs)
__device__ __forceinline__ unsigned bf16_bits(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ void split_bf16x4(const float4& v, uint2& hi, uint2& lo) {
  const unsigned h0 = bf16_bits(v.x), h1 = bf16_bits(v.y), h2 = bf16_bits(v.z), h3 = bf16_bits(v.w);
  const unsigned l0 = bf16_bits(v.x - __uint_as_float(h0 << 16)), l1 = bf16_bits(v.y - __uint_as_float(h1 << 16));
  const unsigned l2 = bf16_bits(v.z - __uint_as_float(h2 << 16)), l3 = bf16_bits(v.w - __uint_as_float(h3 << 16));
  hi = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
  lo = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
}

// f16 hi + lo pieces of four floats (x = hi + lo to 2^-22 |x|; lo pieces of |x| < 2^-3 are subnormal f16: absolute error 2^-25)
__device__ __forceinline__ void split_f16x4(const float4& v, uint2& hi, uint2& lo) {
  f16x4 h, l;
  h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
  l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
  l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}

constexpr int kStageStride = 36;     // floats per staged logits row (32 keys + 4: the 4 lane groups of a store hit disjoint banks)

// One 16-key tile of the relative-position logits: 4 RT MFMAs per 16-channel chunk against the LDS-resident folded queries;
// the tile's (16 RT rows, 16 keys) result (+ the equivariant term) goes to the wave's LDS staging block, columns col0 .. +15.
// NEXT: as soon as the MFMAs of chunk t are issued, b[t] is re-loaded with the same chunk of the wave's next tile (rotating
// prefetch: about one tile of loads stays in flight per wave at all times, so the embedding stream never drains).
// Erow / eq_row are wave-uniform bases; EQ / UA are compile-time because a run-time branch around the e4 loads would put
// their s_waitcnt vmcnt(0) at the top of every tile and drain the prefetched loads.  UA (H % 4 == 0): the 4 rows a lane owns
// in a row tile share one anchor.
// BF: the embedding is stored in bf16.  In 4-byte words the address pattern is the f32 one with half the row stride and half
// the chunks (a lane's 16-byte load = 8 consecutive channels = its B operand of v_mfma_f32_16x16x32_bf16); the folded queries
// are split into bf16 hi + lo fragments (two MFMAs per chunk and row tile: the query side stays exact to 2^-16).
// HS (round 3, the default for the f32 embedding): the f32 MFMAs (1/16 of the f16 rate: 225 us of matrix time per equivariant call at the
// bench shape, as much as the HBM stream itself) are replaced by v_mfma_f32_16x16x32_f16 on f16 hi + lo pieces of BOTH operands -- the folded
// queries split once when they are staged in LDS, the embedding values in registers right behind their load -- three products
// hi hi + hi lo + lo hi in f32 (2^-22 per term: 1e-6 on logits of magnitude 4).  A lane's two float4 of a 32-channel step are not
// consecutive channels; the A fragments are staged with the same permutation of the K index.
// TL (round 4, with HS): the embedding tile is REQUESTED with four adjacent lanes on one key's 64 contiguous bytes (lane l: key l >> 2,
// 16-byte piece l & 3) instead of the MFMA operand's own order (key l & 15, piece l >> 4: the four lanes of a 64-byte segment 16 lanes apart,
// every quarter-wave of a load touching 16 different cache lines), and brought into operand order through a wave-private LDS block right
// before its use (ds_write_b128 / ds_read_b128 of one wave complete in order: no barrier).
template <int CT, int RT, bool NEXT, bool EQ, bool UA, bool BF, bool HS = false, bool TL = false>
__device__ __forceinline__ void bias_tile(float4 (&b)[BF ? CT / 2 : CT], const float4* afrag, const float4* qe_s, float* stage,
                                          int col0, const float* Erow, const float* eq_row, unsigned eq_anchor_stride, int tile,
                                          int next_tile, int M, int AH, int H, float4* tbuf = nullptr, float inv_qs = 1.f) {
  constexpr int CTB = BF ? CT / 2 : CT;      // 16-byte register chunks per tile and lane
  constexpr int CE = CTB * 16;               // embedding row stride in 4-byte words
  constexpr int NF = BF ? 2 * RT : RT;       // A fragments per chunk (bf16: hi and lo)
  const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
  asm volatile("" ::: "memory");        // keeps the LDS fragment reads inside the tile loop (otherwise hoisted and spilled)
  const int m = (tile << 4) + col;
  const float* Enext = TL ? Erow + ((unsigned)min((next_tile << 4) + (lane >> 2), M - 1) * CE + 4 * (lane & 3))
                          : Erow + ((unsigned)min((next_tile << 4) + col, M - 1) * CE + 4 * kq);     // chunk t at +16 t (immediate)
  float4 e4[RT];
  if (EQ && UA) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
      const int a = min((16 * rt + 4 * kq) / H, AH / H - 1);
      e4[rt] = ld4(eq_row + ((unsigned)a * eq_anchor_stride + (unsigned)min(m, M - 1) * 4));
    }
  }
  f32x4 acc[RT], acc_lo[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) acc[rt] = acc_lo[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (HS) {
    constexpr int KT = CT / 2;              // 32-channel steps; fragments: hi at (rt * KT + j) * 64, lo RT * KT * 64 further
#pragma unroll
    for (int j = 0; j < KT; j++) {
      float4 ah[RT], al[RT];              // (not read ahead: the three waves of a SIMD cover the LDS latency, and 16 more registers spill)
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        ah[rt] = afrag[(rt * KT + j) * 64 + lane];
        al[rt] = afrag[((RT + rt) * KT + j) * 64 + lane];
      }
      float4 b0 = b[2 * j], b1 = b[2 * j + 1];
      if (TL) {      // request order (key l >> 2, piece l & 3) -> operand order (key l & 15, piece l >> 4); rows 5 float4 apart (80 B)
        tbuf[(lane >> 2) * 5 + (lane & 3)] = b0;
        tbuf[80 + (lane >> 2) * 5 + (lane & 3)] = b1;
        b0 = tbuf[col * 5 + kq];
        b1 = tbuf[80 + col * 5 + kq];
      }
      uint2 h0, l0, h1, l1;
      split_f16x4(b0, h0, l0);
      split_f16x4(b1, h1, l1);
      const f16x8 bh = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
      const f16x8 bl = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
#pragma unroll
      for (int rt = 0; rt < RT; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al[rt]), bh, acc[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[rt]), bl, acc[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[rt]), bh, acc[rt], 0, 0, 0);
      if (NEXT) {
        b[2 * j] = ld4(Enext + 16 * (2 * j));
        b[2 * j + 1] = ld4(Enext + 16 * (2 * j + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  float4 a[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) a[f] = afrag[(f * CTB) * 64 + lane];
#pragma unroll
  for (int t = 0; t < CTB; t++) {
    // one scheduling group per chunk: [LDS fragments of chunk t+1] [the MFMAs of chunk t] [reload of b[t]]; the barrier keeps
    // the machine scheduler from sinking the reloads behind the last MFMA (which would undo the prefetch)
    float4 an[NF];
    if (t + 1 < CTB) {
#pragma unroll
      for (int f = 0; f < NF; f++) an[f] = afrag[(f * CTB + t + 1) * 64 + lane];
    }
    const float4 bt = b[t];
    if (BF) {
      const bf16x8 bb = __builtin_bit_cast(bf16x8, bt);
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[rt]), bb, acc[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
        acc_lo[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[RT + rt]), bb, acc_lo[rt], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(a[rt], i), f4get(bt, i), acc[rt], 0, 0, 0);
      }
    }
    if (NEXT) b[t] = ld4(Enext + 16 * t);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < CTB) {
#pragma unroll
      for (int f = 0; f < NF; f++) a[f] = an[f];
    }
  }
  }
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int row = 16 * rt + 4 * kq + j;      // rows >= AH: zero queries, never written out
      float val = (BF ? acc[rt][j] + acc_lo[rt][j] : acc[rt][j]) * inv_qs;      // (the folded queries were scaled by a power of two before their split)
      if (EQ) {
        const float4 e = UA ? e4[rt]
                            : ld4(eq_row + ((unsigned)min(row / H, AH / H - 1) * eq_anchor_stride + (unsigned)min(m, M - 1) * 4));
        const float4 w = qe_s[row];
        val += (w.x * e.x + w.y * e.y) + (w.z * e.z + w.w * e.w);
      }
      stage[row * kStageStride + col0 + col] = m < M ? val : 0.f;
    }
  }
}

// Flat, balanced decomposition: the (row n, 32-key unit) pairs of all clouds form one list; workgroup w owns the contiguous
// range [w U / G, (w+1) U / G) and walks it row segment by row segment (the folded queries of the segment's row are staged
// in LDS in MFMA-fragment order); inside a segment the 4 waves stride over the units.  A unit = two 16-key tiles whose
// logits are collected in a wave-private LDS block and written out as full 128-byte row segments (float4 per lane).
template <int CT, int RT, int MINW, bool EQ, bool UA, bool BF = false, bool HS = false, bool TL = false>
__global__ __launch_bounds__(256, MINW) void rpe_bias_kernel(const float* __restrict__ qp, const float* __restrict__ qe,
                                                             int qp_rs, long long qp_sa, Stack S, int AH, int H,
                                                             float* __restrict__ bias) {
  constexpr int C = CT * 16;
  constexpr int CTB = BF ? CT / 2 : CT, CE = CTB * 16;      // 16-byte chunks per tile and lane, row stride in 4-byte words
  __shared__ float4 afrag[RT * CT * 64];                    // bf16: hi fragments, then lo fragments (same size)
  __shared__ float4 qe_s[32];
  __shared__ __attribute__((aligned(16))) float stage_s[4 * RT * 16 * kStageStride];
  __shared__ float4 tbuf_s[TL ? 4 * 160 : 1];                 // TL: per wave two 16-row blocks of 80 bytes per row
  // Round 5, late (the logits kernel's half): the folded queries of a query point are scaled by ONE power of two
  // before their f16 / bf16 split -- the largest magnitude over the point's (anchor, head) rows goes to [2^6, 2^7) unless it already lies in
  // [2^-4, 2^7) (then nothing changes: bit-identical) -- and the logits are scaled back in the tile epilogue: no overflow to Inf above 65504,
  // no loss of the lo piece below 2^-3.  The four wave maxima travel through LDS in front of the barrier the staging has anyway.
  __shared__ unsigned qmax_s[2][4];
  int seg_parity = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
  float* stage = stage_s + wave * (RT * 16 * kStageStride);
  float4* tbuf = tbuf_s + (TL ? wave * 160 : 0);
  int f = (int)((long long)blockIdx.x * S.total_units / gridDim.x);
  const int f_end = (int)((long long)(blockIdx.x + 1) * S.total_units / gridDim.x);
  while (f < f_end) {
    int ci = 0;
    for (int i = 1; i < S.n; i++)
      if (f >= S.c[i].unit_begin) ci = i;
    const StackCloud cl = stack_pick(S, ci);
    const int units = cl.Mp >> 5;
    const int rel = f - cl.unit_begin;
    const int n = rel / units, u_lo = rel - n * units;
    const int u_hi = min(units, u_lo + (f_end - f));
    f += u_hi - u_lo;

    const float* Erow = cl.emb + (size_t)n * cl.M * CE;          // wave-uniform base of the embedding row block
    int unit = u_lo + wave;
    // Issue order (the VMEM counter is in-order): folded queries of packed row q_start + n first, then the wave's first
    // embedding tile; the queries are waited for and written to LDS while the tile is still in flight.
    // afrag[(rt*CT + t)*64 + kq*16 + r] = float4 of channels 16 t + 4 kq .. +3 of folded query row 16 rt + r (r = a*H + h);
    // thread i takes row i % (16 RT) and chunk i / (16 RT): consecutive lanes write consecutive float4 (conflict free).
    constexpr int QTOT = RT * 16 * (C / 4), QI = (QTOT + 255) / 256;
    const size_t qrow = (size_t)(cl.q_start + n) * qp_rs;
    float4 qv[QI];
#pragma unroll
    for (int u = 0; u < QI; u++) {
      const int i = threadIdx.x + 256 * u;
      const int row = i % (RT * 16), c4 = i / (RT * 16);
      qv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < QTOT && row < AH) {
        const int a = row / H, h = row - a * H;
        qv[u] = ld4(qp + (size_t)a * qp_sa + qrow + h * C + 4 * c4);
      }
    }
    float4 qev = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EQ && threadIdx.x < AH) {
      const int a = threadIdx.x / H, h = threadIdx.x - a * H;
      qev = ld4(qe + (size_t)a * qp_sa + qrow + 4 * h);
    }
    float4 b[CTB];
    if (unit < u_hi) {
      const float* E0 = TL ? Erow + ((unsigned)min((unit << 5) + (lane >> 2), cl.M - 1) * CE + 4 * (lane & 3))
                           : Erow + ((unsigned)min((unit << 5) + col, cl.M - 1) * CE + 4 * kq);
#pragma unroll
      for (int t = 0; t < CTB; t++) b[t] = ld4(E0 + 16 * t);
    }
    float qs = 1.f, inv_qs = 1.f;
    if (BF || HS) {
      unsigned mb = 0;                     // (non-negative floats order like their bit patterns; NaN / Inf end up on top: no scaling)
#pragma unroll
      for (int u = 0; u < QI; u++) {
        mb = max(mb, __float_as_uint(fabsf(qv[u].x)));
        mb = max(mb, __float_as_uint(fabsf(qv[u].y)));
        mb = max(mb, __float_as_uint(fabsf(qv[u].z)));
        mb = max(mb, __float_as_uint(fabsf(qv[u].w)));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
      if (lane == 0) qmax_s[seg_parity][wave] = mb;
    }
    __syncthreads();                      // the previous segment's fragment reads are done
    if (BF || HS) {
      const unsigned mb = max(max(qmax_s[seg_parity][0], qmax_s[seg_parity][1]), max(qmax_s[seg_parity][2], qmax_s[seg_parity][3]));
      seg_parity ^= 1;
      const int e = (int)((mb >> 23) & 0xff);
      if (!(e == 0 || e == 0xff || (e >= 123 && e <= 133))) {
        qs = __uint_as_float((unsigned)(127 + 133 - e) << 23);
        inv_qs = __uint_as_float((unsigned)(127 - 133 + e) << 23);
#pragma unroll
        for (int u = 0; u < QI; u++) qv[u] = make_float4(qv[u].x * qs, qv[u].y * qs, qv[u].z * qs, qv[u].w * qs);
      }
    }
#pragma unroll
    for (int u = 0; u < QI; u++) {
      const int i = threadIdx.x + 256 * u;
      const int row = i % (RT * 16), c4 = i / (RT * 16);
      if (BF) {
        // bf16 fragment (rt, t32)[kq * 16 + r] = channels 32 t32 + 8 kq .. +7 of row 16 rt + r: this float4 is one 8-byte half
        const int t32 = c4 >> 3, kk = (c4 >> 1) & 3, hf = c4 & 1;
        uint2 hi, lo;
        split_bf16x4(qv[u], hi, lo);
        uint2* af2 = reinterpret_cast<uint2*>(afrag);
        const int slot = ((((row >> 4) * CTB + t32) * 64 + kk * 16 + (row & 15)) << 1) + hf;
        if (i < QTOT) {
          af2[slot] = hi;
          af2[slot + RT * CTB * 64 * 2] = lo;
        }
      } else if (HS) {
        // f16 fragment (rt, j)[kk * 16 + r] = channels 32 j + 4 kk .. +3 and 32 j + 16 + 4 kk .. +3 of row 16 rt + r (the K order of a
        // lane's two embedding float4 of that step): this float4 is one 8-byte half
        const int t = c4 >> 2, kk = c4 & 3, j = t >> 1, hf = t & 1;
        uint2 hi, lo;
        split_f16x4(qv[u], hi, lo);
        uint2* af2 = reinterpret_cast<uint2*>(afrag);
        const int slot = ((((row >> 4) * (CT / 2) + j) * 64 + kk * 16 + (row & 15)) << 1) + hf;
        if (i < QTOT) {
          af2[slot] = hi;
          af2[slot + RT * (CT / 2) * 64 * 2] = lo;
        }
      } else {
        const int t = c4 >> 2, kk = c4 & 3;
        if (i < QTOT) afrag[((row >> 4) * CT + t) * 64 + kk * 16 + (row & 15)] = qv[u];
      }
    }
    if (threadIdx.x < 32) qe_s[threadIdx.x] = qev;
    __syncthreads();

    const unsigned eq_sa = (unsigned)cl.N * cl.M * 4;
    const float* eq_row = EQ ? cl.eq + (size_t)n * cl.M * 4 : nullptr;
    float* bias_row = bias + cl.bias_off + (size_t)n * cl.Mp;
    const unsigned bias_ah = (unsigned)cl.N * cl.Mp;
    for (; unit < u_hi; unit += 4) {
      const int t0 = unit << 1;
      bias_tile<CT, RT, true, EQ, UA, BF, HS, TL>(b, afrag, qe_s, stage, 0, Erow, eq_row, eq_sa, t0, t0 + 1, cl.M, AH, H, tbuf, inv_qs);
      if (unit + 4 < u_hi)
        bias_tile<CT, RT, true, EQ, UA, BF, HS, TL>(b, afrag, qe_s, stage, 16, Erow, eq_row, eq_sa, t0 + 1, t0 + 8, cl.M, AH, H, tbuf, inv_qs);
      else
        bias_tile<CT, RT, false, EQ, UA, BF, HS, TL>(b, afrag, qe_s, stage, 16, Erow, eq_row, eq_sa, t0 + 1, t0 + 1, cl.M, AH, H, tbuf, inv_qs);
      // the unit's (rows, 32 keys) block: 8 lanes cover one row's 128 bytes
      const int r8 = lane >> 3, m4 = (lane & 7) * 4;
#pragma unroll
      for (int pass = 0; pass < RT * 2; pass++) {
        const int row = pass * 8 + r8;
        if (row < AH) {
          const float4 v = *reinterpret_cast<const float4*>(stage + row * kStageStride + m4);
          *reinterpret_cast<float4*>(bias_row + ((unsigned)row * bias_ah + (unsigned)((unit << 5) + m4))) = v;
        }
      }
    }
  }
}

// =====================================================================================================================
// flash-style core: one wave, one 32-query tile, a strided set of 32-key tiles
// =====================================================================================================================
template <int D>
struct FlashState {
  static constexpr int DT = (D + 31) / 32;
  static constexpr int KU = D / 8;
  float m, l;
  f32x16 o[DT];
};

template <int D>
struct TileRegs {
  float4 kf[FlashState<D>::KU];
  float4 b4[4];
  float vv[FlashState<D>::DT * 16];
};

template <int D>
__device__ __forceinline__ void flash_load(TileRegs<D>& r, const float* __restrict__ k, const float* __restrict__ v,
                                           const float* __restrict__ bias_row, int m0, int M, int k_rs, int v_rs) {
  constexpr int DT = FlashState<D>::DT, KU = FlashState<D>::KU;
  const int lane = threadIdx.x & 63, half = lane >> 5, c32 = lane & 31;
  const float* kr = k + (size_t)min(m0 + c32, M - 1) * k_rs + 4 * half;
#pragma unroll
  for (int u = 0; u < KU; u++) r.kf[u] = ld4(kr + 8 * u);
#pragma unroll
  for (int g = 0; g < 4; g++)
    r.b4[g] = bias_row ? ld4(bias_row + m0 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int dt = 0; dt < DT; dt++) {
    const int dd = 32 * dt + c32;
#pragma unroll
    for (int g = 0; g < 4; g++) {          // keys m0 + 8 g + 4 half .. +3 are contiguous in the transposed values
      const float4 t = dd < D ? ld4(v + (size_t)dd * v_rs + m0 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
      r.vv[dt * 16 + 4 * g + 0] = t.x;
      r.vv[dt * 16 + 4 * g + 1] = t.y;
      r.vv[dt * 16 + 4 * g + 2] = t.z;
      r.vv[dt * 16 + 4 * g + 3] = t.w;
    }
  }
}

// q/k point at the (anchor, head) slice (row strides q_rs / k_rs floats: they may be column blocks of a wider projection); v points at the TRANSPOSED values of the slice, vt[dd * Mp + key]
// (keys zero-padded to Mp, a multiple of 32).  bias (may be null) points at the (ah) slice, row stride Mp.
// PREFETCH: the loads of tile t+step are issued before the MFMAs of tile t (two register sets); otherwise the loads of a tile
// sit at the top of its iteration and latency is hidden by the other resident waves (fewer VGPRs, more waves per SIMD).
template <int D, bool PREFETCH = true>
__device__ __forceinline__ void flash_tiles(FlashState<D>& st, const float* __restrict__ q, const float* __restrict__ k,
                                            const float* __restrict__ v, const float* __restrict__ bias, int n0, int N, int M,
                                            int q_rs, int k_rs, int v_rs, int Mp, float scale, int tile_begin, int tile_step) {
  constexpr int DT = FlashState<D>::DT, KU = FlashState<D>::KU;
  const int lane = threadIdx.x & 63, half = lane >> 5, c32 = lane & 31;
  const int nq = min(n0 + c32, N - 1);
  float4 qf[KU];
#pragma unroll
  for (int u = 0; u < KU; u++) qf[u] = ld4(q + (size_t)nq * q_rs + 8 * u + 4 * half);
  const float* bias_row = bias ? bias + (size_t)nq * Mp : nullptr;
  const int tiles = (M + 31) >> 5;
  TileRegs<D> cur, nxt;
  int tile = tile_begin;
  if (PREFETCH && tile < tiles) flash_load<D>(cur, k, v, bias_row, tile << 5, M, k_rs, v_rs);
  for (; tile < tiles; tile += tile_step) {
    const int m0 = tile << 5;
    const bool more = PREFETCH && tile + tile_step < tiles;
    if (PREFETCH) {
      if (more) flash_load<D>(nxt, k, v, bias_row, (tile + tile_step) << 5, M, k_rs, v_rs);
    } else {
      flash_load<D>(cur, k, v, bias_row, m0, M, k_rs, v_rs);
    }
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.f;
#pragma unroll
    for (int u = 0; u < KU; u++)
#pragma unroll
      for (int i = 0; i < 4; i++)
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(cur.kf[u], i), f4get(qf[u], i), s, 0, 0, 0);
    // s[r] = S^T[key = (r&3) + 8 (r>>2) + 4 half][query = c32]
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int key = m0 + 8 * g + 4 * half + j;
        float val = (s[4 * g + j] + f4get(cur.b4[g], j)) * scale;
        val = key < M ? val : -INFINITY;
        s[4 * g + j] = val;
        mx = fmaxf(mx, val);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(st.m, mx);
    const float alpha = __expf(st.m - m_new);        // st.m = -inf on the first tile -> 0
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      s[r] = __expf(s[r] - m_new);
      ps += s[r];
    }
    ps += __shfl_xor(ps, 32);
    st.l = st.l * alpha + ps;
    st.m = m_new;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
#pragma unroll
      for (int r = 0; r < 16; r++) st.o[dt][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; r++) st.o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.vv[dt * 16 + r], s[r], st.o[dt], 0, 0, 0);
    }
    if (PREFETCH && more) cur = nxt;
  }
}

// Variant of flash_tiles for the attention kernel: the workgroup's 32-query tile sits in LDS (q_s[(2 u + half) * 32 + c32] =
// float4 q[n0 + c32][8 u + 4 half ..], shared by the waves, conflict-free ds_read_b128); every operand of a key tile is requested
// one phase or one tile ahead of its use, each into registers that have just been consumed (no second register set, no spills
// at 3 workgroups = 12 waves per CU; a single scratch reload would cost a vmcnt(0) drain in the middle of the tile).
#define SE3_STAMP(slot)                                                    \
  if (PROF) {                                                              \
    __builtin_amdgcn_sched_barrier(0);                                     \
    if (prof != nullptr && (threadIdx.x & 63) == 0) prof[slot] = clock64(); \
    __builtin_amdgcn_sched_barrier(0);                                     \
  }
template <int D, bool HAS_BIAS, bool PROF = false>
__device__ __forceinline__ void flash_tiles_qlds(FlashState<D>& st, const float4* q_s, const float* __restrict__ k,
                                                 const float* __restrict__ v, const float* __restrict__ bias, int n0, int N, int M,
                                                 int k_rs, int v_rs, int Mp, float scale, int tile_begin, int tile_step,
                                                 long long* prof = nullptr) {
  constexpr int DT = FlashState<D>::DT, KU = FlashState<D>::KU;
  const int lane = threadIdx.x & 63, half = lane >> 5, c32 = lane & 31;
  const int nq = min(n0 + c32, N - 1);
  const float* bias_row = HAS_BIAS ? bias + (size_t)nq * Mp : nullptr;
  const int tiles = (M + 31) >> 5;
  if (tile_begin >= tiles) return;
  float4 kf[KU], b4[4];
  {
    const float* kr = k + (size_t)min((tile_begin << 5) + c32, M - 1) * k_rs + 4 * half;
#pragma unroll
    for (int u = 0; u < KU; u++) kf[u] = ld4(kr + 8 * u);
#pragma unroll
    for (int g = 0; g < 4; g++)
      b4[g] = HAS_BIAS ? ld4(bias_row + (tile_begin << 5) + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int pslot = 2;
  for (int tile = tile_begin; tile < tiles; tile += tile_step) {
    const int m0 = tile << 5;
    const int tn = min(tile + tile_step, tiles - 1);      // the last iteration re-requests its own tile: branch-free
    float vv[DT * 16];
    SE3_STAMP(pslot)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.f;
    float4 qf = q_s[half * 32 + c32];
#pragma unroll
    for (int u = 0; u < KU; u++) {
      // the LDS read of fragment u+1 is issued in front of the 4 dependent MFMAs of fragment u (in-order issue: behind
      // them it would expose the LDS latency once per fragment)
      float4 qn;
      if (u + 1 < KU) qn = q_s[(2 * (u + 1) + half) * 32 + c32];
#pragma unroll
      for (int i = 0; i < 4; i++) s = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(kf[u], i), f4get(qf, i), s, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 1 < KU) qf = qn;
    }
    // (1) V^T of this tile, (2) K fragment of the next tile -- straight into the registers the S MFMAs just read.  The
    // compiler fences keep the requests here (values used in the next iteration are otherwise sunk to the loop latch, i.e.
    // issued right in front of their first use).
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      const int dd = 32 * dt + c32;
#pragma unroll
      for (int g = 0; g < 4; g++) {          // keys m0 + 8 g + 4 half .. +3 are contiguous in the transposed values
        const float4 t = dd < D ? ld4(v + (size_t)dd * v_rs + m0 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        vv[dt * 16 + 4 * g + 0] = t.x;
        vv[dt * 16 + 4 * g + 1] = t.y;
        vv[dt * 16 + 4 * g + 2] = t.z;
        vv[dt * 16 + 4 * g + 3] = t.w;
      }
    }
    {
      const float* kr = k + (size_t)min((tn << 5) + c32, M - 1) * k_rs + 4 * half;
#pragma unroll
      for (int u = 0; u < KU; u++) kf[u] = ld4(kr + 8 * u);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // s[r] = S^T[key = (r&3) + 8 (r>>2) + 4 half][query = c32]
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int key = m0 + 8 * g + 4 * half + j;
        float val = (s[4 * g + j] + f4get(b4[g], j)) * scale;
        val = key < M ? val : -INFINITY;
        s[4 * g + j] = val;
        mx = fmaxf(mx, val);
      }
    }
    // (3) the logits of the next tile, into the registers just consumed.  They were written by the previous kernel on other
    // XCDs and come from the memory side (about two S phases away): requested a full tile ahead, and last, because the
    // VMEM counter is in-order -- the P.V phase below waits for V^T only.
    __builtin_amdgcn_sched_barrier(0);
    if (HAS_BIAS) {
#pragma unroll
      for (int g = 0; g < 4; g++) b4[g] = ld4(bias_row + (tn << 5) + 8 * g + 4 * half);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    SE3_STAMP(pslot + 1)
    const float m_new = fmaxf(st.m, mx);
    const float alpha = __expf(st.m - m_new);        // st.m = -inf on the first tile -> 0
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      s[r] = __expf(s[r] - m_new);
      ps += s[r];
    }
    ps += __shfl_xor(ps, 32);
    st.l = st.l * alpha + ps;
    st.m = m_new;
    SE3_STAMP(pslot + 2)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
#pragma unroll
      for (int r = 0; r < 16; r++) st.o[dt][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; r++) st.o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[dt * 16 + r], s[r], st.o[dt], 0, 0, 0);
    }
    SE3_STAMP(pslot + 3)
    pslot += 4;
  }
}

// f16-split form of flash_tiles_qlds (round 3; head dimensions that are multiples of 16): both products on v_mfma_f32_32x32x16_f16 with every
// operand as f16 hi + lo pieces and three products hi hi + hi lo + lo hi accumulated in f32 (2^-22 per term) -- 24 MFMAs of 32 cycles per
// (32 query, 32 key) tile where the f32 form issues 64 of 64 cycles (93 us of matrix time per equivariant call at the bench shape).  The
// query tile is split once into LDS (q_h / q_l[(2 j + half) * 32 + c32] = 8 f16: channels 16 j + 4 half .. +3 and 16 j + 8 + 4 half .. +3, the
// order in which a lane's two K float4 of MFMA j arrive); K, V^T and the softmax weights are split in registers.  The S^T accumulator
// registers of a lane are keys {8 g + 4 half + j}: registers 0..7 / 8..15 are directly the B operand of the two K16 steps of P V, and the V^T
// registers are loaded in the same key order.
__device__ __forceinline__ void split_f16x8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    hi[i] = (_Float16)v[i];
    lo[i] = (_Float16)(v[i] - (float)hi[i]);
  }
}
template <int D, bool HAS_BIAS>
__device__ __forceinline__ void flash_tiles_f16(FlashState<D>& st, const uint4* q_h, const uint4* q_l, const float* __restrict__ k,
                                                const float* __restrict__ v, const float* __restrict__ bias, int n0, int N, int M,
                                                int k_rs, int v_rs, int Mp, float scale, int tile_begin, int tile_step) {
  constexpr int DT = FlashState<D>::DT, KU = FlashState<D>::KU, KJ = D / 16;
  static_assert(D % 16 == 0, "f16 flash tiles need a head dimension that is a multiple of 16");
  const int lane = threadIdx.x & 63, half = lane >> 5, c32 = lane & 31;
  const int nq = min(n0 + c32, N - 1);
  const float* bias_row = HAS_BIAS ? bias + (size_t)nq * Mp : nullptr;
  const int tiles = (M + 31) >> 5;
  if (tile_begin >= tiles) return;
  float4 kf[KU], b4[4];
  {
    const float* kr = k + (size_t)min((tile_begin << 5) + c32, M - 1) * k_rs + 4 * half;
#pragma unroll
    for (int u = 0; u < KU; u++) kf[u] = ld4(kr + 8 * u);
#pragma unroll
    for (int g = 0; g < 4; g++)
      b4[g] = HAS_BIAS ? ld4(bias_row + (tile_begin << 5) + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int tile = tile_begin; tile < tiles; tile += tile_step) {
    const int m0 = tile << 5;
    const int tn = min(tile + tile_step, tiles - 1);      // the last iteration re-requests its own tile: branch-free
    float vv[DT * 16];
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < KJ; j++) {
      const f16x8 qh = __builtin_bit_cast(f16x8, q_h[(2 * j + half) * 32 + c32]);
      const f16x8 ql = __builtin_bit_cast(f16x8, q_l[(2 * j + half) * 32 + c32]);
      const float kv[8] = {kf[2 * j].x, kf[2 * j].y, kf[2 * j].z, kf[2 * j].w, kf[2 * j + 1].x, kf[2 * j + 1].y, kf[2 * j + 1].z, kf[2 * j + 1].w};
      f16x8 kh, kl;
      split_f16x8(kv, kh, kl);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh, s, 0, 0, 0);
    }
    // (1) V^T of this tile, (2) K fragment of the next tile -- into the registers the S phase just read (see flash_tiles_qlds)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      const int dd = 32 * dt + c32;
#pragma unroll
      for (int g = 0; g < 4; g++) {          // keys m0 + 8 g + 4 half .. +3 are contiguous in the transposed values
        const float4 t = dd < D ? ld4(v + (size_t)dd * v_rs + m0 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        vv[dt * 16 + 4 * g + 0] = t.x;
        vv[dt * 16 + 4 * g + 1] = t.y;
        vv[dt * 16 + 4 * g + 2] = t.z;
        vv[dt * 16 + 4 * g + 3] = t.w;
      }
    }
    {
      const float* kr = k + (size_t)min((tn << 5) + c32, M - 1) * k_rs + 4 * half;
#pragma unroll
      for (int u = 0; u < KU; u++) kf[u] = ld4(kr + 8 * u);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int key = m0 + 8 * g + 4 * half + j;
        float val = (s[4 * g + j] + f4get(b4[g], j)) * scale;
        val = key < M ? val : -INFINITY;
        s[4 * g + j] = val;
        mx = fmaxf(mx, val);
      }
    }
    // (3) the logits of the next tile, into the registers just consumed, requested last (in-order VMEM counter)
    __builtin_amdgcn_sched_barrier(0);
    if (HAS_BIAS) {
#pragma unroll
      for (int g = 0; g < 4; g++) b4[g] = ld4(bias_row + (tn << 5) + 8 * g + 4 * half);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(st.m, mx);
    const float alpha = __expf(st.m - m_new);        // st.m = -inf on the first tile -> 0
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      s[r] = __expf(s[r] - m_new);
      ps += s[r];
    }
    ps += __shfl_xor(ps, 32);
    st.l = st.l * alpha + ps;
    st.m = m_new;
    f16x8 ph[2], pl[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float pv[8] = {s[8 * i], s[8 * i + 1], s[8 * i + 2], s[8 * i + 3], s[8 * i + 4], s[8 * i + 5], s[8 * i + 6], s[8 * i + 7]};
      split_f16x8(pv, ph[i], pl[i]);
    }
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
#pragma unroll
      for (int r = 0; r < 16; r++) st.o[dt][r] *= alpha;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const float vs[8] = {vv[dt * 16 + 8 * i], vv[dt * 16 + 8 * i + 1], vv[dt * 16 + 8 * i + 2], vv[dt * 16 + 8 * i + 3],
                             vv[dt * 16 + 8 * i + 4], vv[dt * 16 + 8 * i + 5], vv[dt * 16 + 8 * i + 6], vv[dt * 16 + 8 * i + 7]};
        f16x8 vh, vl;
        split_f16x8(vs, vh, vl);
        st.o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[i], st.o[dt], 0, 0, 0);
        st.o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[i], st.o[dt], 0, 0, 0);
        st.o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[i], st.o[dt], 0, 0, 0);
      }
    }
  }
}

template <int D>
__device__ __forceinline__ void flash_init(FlashState<D>& st) {
  st.m = -INFINITY;
  st.l = 0.f;
#pragma unroll
  for (int dt = 0; dt < FlashState<D>::DT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) st.o[dt][r] = 0.f;
}

// merge the states of the NW waves of a workgroup into wave 0 (through LDS); returns normalised output in st (wave 0 only)
template <int D, int NW>
__device__ __forceinline__ void flash_merge(FlashState<D>& st, float* sm, float* sl, float* so) {
  constexpr int DT = FlashState<D>::DT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sm[wave * 64 + lane] = st.m;
  sl[wave * 64 + lane] = st.l;
#pragma unroll
  for (int dt = 0; dt < DT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) so[((wave * DT + dt) * 16 + r) * 64 + lane] = st.o[dt][r];
  __syncthreads();
  if (wave == 0) {
    float mt = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; w++) mt = fmaxf(mt, sm[w * 64 + lane]);
    float lt = 0.f;
    float f[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
      f[w] = __expf(sm[w * 64 + lane] - mt);
      lt += sl[w * 64 + lane] * f[w];
    }
    const float inv = 1.f / lt;
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) acc += so[((w * DT + dt) * 16 + r) * 64 + lane] * f[w];
        st.o[dt][r] = acc * inv;
      }
    st.m = mt;
    st.l = lt;
  }
}

// Merge the states of the NW waves through LDS and write the normalised output tile; every wave reduces and stores its own
// share of the output columns (groups of 4 consecutive channels), so the tail of the workgroup is NW-way parallel.
template <int D, int NW>
__device__ __forceinline__ void flash_merge_store(const FlashState<D>& st, float* sm, float* sl, float* so,
                                                  float* __restrict__ out, int n0, int N, int C) {
  constexpr int DT = FlashState<D>::DT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, c32 = lane & 31;
  sm[wave * 64 + lane] = st.m;
  sl[wave * 64 + lane] = st.l;
#pragma unroll
  for (int dt = 0; dt < DT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) so[((wave * DT + dt) * 16 + r) * 64 + lane] = st.o[dt][r];
  __syncthreads();
  float mt = -INFINITY;
#pragma unroll
  for (int w = 0; w < NW; w++) mt = fmaxf(mt, sm[w * 64 + lane]);
  float lt = 0.f;
  float f[NW];
#pragma unroll
  for (int w = 0; w < NW; w++) {
    f[w] = __expf(sm[w * 64 + lane] - mt);
    lt += sl[w * 64 + lane] * f[w];
  }
  const float inv = 1.f / lt;
  const int nq = n0 + c32;
  for (int gi = wave; gi < DT * 4; gi += NW) {          // register group gi = 4 consecutive channels 32 dt + 8 g + 4 half ..
    const int dt = gi >> 2, g = gi & 3;
    const int dd = 32 * dt + 8 * g + 4 * half;
    float val[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) acc += so[((w * DT + dt) * 16 + 4 * g + j) * 64 + lane] * f[w];
      val[j] = acc * inv;
    }
    if (nq < N && dd < D) *reinterpret_cast<float4*>(out + (size_t)nq * C + dd) = make_float4(val[0], val[1], val[2], val[3]);
  }
}

template <int D>
__device__ __forceinline__ void flash_store(const FlashState<D>& st, float* __restrict__ out, int n0, int N, int C,
                                            float weight, bool accumulate) {
  constexpr int DT = FlashState<D>::DT;
  const int lane = threadIdx.x & 63, half = lane >> 5, c32 = lane & 31;
  const int nq = n0 + c32;
  if (nq >= N) return;
#pragma unroll
  for (int dt = 0; dt < DT; dt++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int dd = 32 * dt + 8 * g + 4 * half;
      if (dd < D) {
        float4* p = reinterpret_cast<float4*>(out + (size_t)nq * C + dd);
        float4 val = make_float4(st.o[dt][4 * g] * weight, st.o[dt][4 * g + 1] * weight, st.o[dt][4 * g + 2] * weight,
                                 st.o[dt][4 * g + 3] * weight);
        if (accumulate) {
          const float4 old = *p;
          val = make_float4(val.x + old.x, val.y + old.y, val.z + old.z, val.w + old.w);
        }
        *p = val;
      }
    }
}

struct AttnArgs {
  const float *q, *k, *v, *bias;
  float* out;
  Stack S;
  int C, H, A;
  int q_rs, k_rs, v_rs;               // row strides of q, k (>= C) and of the transposed values in floats
  long long q_sa, k_sa, v_sa, o_sa;   // anchor strides in floats (0 = shared by all anchors)
  float scale;
  int QT;                             // 32-query tiles of the largest cloud
  int G;                              // groups = clouds * anchors * heads
  long long* prof;                    // profiling hook (attention variant 9): 32 stamps per wave, else null
};

// One workgroup per (cloud, anchor, head, 32-query tile); its NW waves split the key tiles and merge through LDS.
// XCD-aware flat grid: workgroup i runs on XCD i % 8 (round-robin dispatch), so group g = (cloud, anchor, head) is pinned
// to XCD g % 8 and the K / V^T of a group (re-read by all of its query tiles) stay in ONE L2.
template <int D, int NW, int MINW, int MODE>      // MODE 0: loads at the tile top; 1: full double buffering; 2: Q in LDS + K prefetch
__global__ __launch_bounds__(64 * NW, MINW) void attention_kernel(AttnArgs p) {
  __shared__ float sm[64 * NW], sl[64 * NW];
  __shared__ float so[NW * FlashState<D>::DT * 16 * 64];
  __shared__ float4 q_s[MODE >= 2 ? (D / 8) * 2 * 32 : 1];
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int gslot = j / p.QT, qt = j - gslot * p.QT;
  const int g = gslot * 8 + xcd;
  if (g >= p.G) return;
  const int AH = p.A * p.H;
  const int ci = g / AH, ah = g - ci * AH;
  const int a = ah / p.H, h = ah - a * p.H;
  const StackCloud cl = stack_pick(p.S, ci);
  const int n0 = qt * 32;
  if (n0 >= cl.N) return;
  const int wave = threadIdx.x >> 6;
  FlashState<D> st;
  flash_init(st);
  const float* bias = p.bias ? p.bias + cl.bias_off + ((size_t)ah * cl.N) * cl.Mp : nullptr;
  const float* q = p.q + a * p.q_sa + (size_t)cl.q_start * p.q_rs + h * D;
  const float* k = p.k + a * p.k_sa + (size_t)cl.k_start * p.k_rs + h * D;
  const float* v = p.v + a * p.v_sa + (size_t)h * D * p.v_rs + cl.k_start;
  long long* prof = nullptr;         // MODE 3 (profiling hook): 32 clock64() stamps per wave
  if (MODE == 3 && p.prof != nullptr) {
    prof = p.prof + ((size_t)blockIdx.x * NW + wave) * 32;
    if ((threadIdx.x & 63) == 0) prof[0] = clock64();
  }
  if (MODE == 4) {
    // the query tile as f16 hi / lo pieces: fragment (j, half)[c32] = channels 16 j + 4 half .. +3 | 16 j + 8 + 4 half .. +3; thread i stages
    // one float4 = one 8-byte half of a fragment
    if constexpr (D % 16 == 0) {
      uint2* qh2 = reinterpret_cast<uint2*>(q_s);
      uint2* ql2 = qh2 + (D / 16) * 2 * 32 * 2;
      for (int i = threadIdx.x; i < (D / 8) * 2 * 32; i += 64 * NW) {
        const int c32 = i & 31, uh = i >> 5, u = uh >> 1, hf = uh & 1;
        const float4 qv = ld4(q + (size_t)min(n0 + c32, cl.N - 1) * p.q_rs + 8 * u + 4 * hf);
        uint2 hi, lo;
        split_f16x4(qv, hi, lo);
        const int slot = ((((u >> 1) * 2 + hf) * 32 + c32) << 1) + (u & 1);
        qh2[slot] = hi;
        ql2[slot] = lo;
      }
      __syncthreads();
      const uint4* q_h = reinterpret_cast<const uint4*>(q_s);
      const uint4* q_l = q_h + (D / 16) * 2 * 32;
      if (p.bias != nullptr)
        flash_tiles_f16<D, true>(st, q_h, q_l, k, v, bias, n0, cl.N, cl.M, p.k_rs, p.v_rs, cl.Mp, p.scale, wave, NW);
      else
        flash_tiles_f16<D, false>(st, q_h, q_l, k, v, bias, n0, cl.N, cl.M, p.k_rs, p.v_rs, cl.Mp, p.scale, wave, NW);
    }
  } else if (MODE >= 2) {
    for (int i = threadIdx.x; i < (D / 8) * 2 * 32; i += 64 * NW) {
      const int c32 = i & 31, uh = i >> 5;
      q_s[i] = ld4(q + (size_t)min(n0 + c32, cl.N - 1) * p.q_rs + 8 * (uh >> 1) + 4 * (uh & 1));
    }
    __syncthreads();
    if (MODE == 3 && prof != nullptr && (threadIdx.x & 63) == 0) prof[1] = clock64();
    if (p.bias != nullptr)
      flash_tiles_qlds<D, true, MODE == 3>(st, q_s, k, v, bias, n0, cl.N, cl.M, p.k_rs, p.v_rs, cl.Mp, p.scale, wave, NW, prof);
    else
      flash_tiles_qlds<D, false, MODE == 3>(st, q_s, k, v, bias, n0, cl.N, cl.M, p.k_rs, p.v_rs, cl.Mp, p.scale, wave, NW, prof);
    if (MODE == 3 && prof != nullptr && (threadIdx.x & 63) == 0) prof[29] = clock64();
  } else {
    flash_tiles<D, MODE == 1>(st, q, k, v, bias, n0, cl.N, cl.M, p.q_rs, p.k_rs, p.v_rs, cl.Mp, p.scale, wave, NW);
  }
  if (MODE == 3 && prof != nullptr && (threadIdx.x & 63) == 0) prof[30] = clock64();
  flash_merge_store<D, NW>(st, sm, sl, so, p.out + a * p.o_sa + (size_t)cl.q_start * p.C + h * D, n0, cl.N, p.C);
  if (MODE == 3 && prof != nullptr && (threadIdx.x & 63) == 0) prof[31] = clock64();
}

// ---------------------------------------------------------------------------------------------------------------------
// equivariant cross attention
// ---------------------------------------------------------------------------------------------------------------------
// grid (ceil(N/32), A*A): partial[blockIdx] = sum over the tile's (n, m) of (mean_h S[a,e,h,n,m])^2
template <int D>
__global__ __launch_bounds__(256) void cross_eq_stats_kernel(const float* __restrict__ q, const float* __restrict__ k, int A,
                                                             int N, int M, int C, int H, float scale,
                                                             float* __restrict__ partial) {
  constexpr int KU = D / 8;
  __shared__ float red[4];
  const int n0 = blockIdx.x * 32, ae = blockIdx.y, a = ae / A, e = ae - a * A;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, c32 = lane & 31;
  const int nq = min(n0 + c32, N - 1);
  const int tiles = (M + 31) >> 5;
  float total = 0.f;
  for (int tile = wave; tile < tiles; tile += 4) {
    const int m0 = tile << 5;
    const int kr = min(m0 + c32, M - 1);
    f32x16 mean;
#pragma unroll
    for (int r = 0; r < 16; r++) mean[r] = 0.f;
    for (int h = 0; h < H; h++) {
      const float* qh = q + ((size_t)a * N + nq) * C + h * D;
      const float* kh = k + ((size_t)e * M + kr) * C + h * D;
#pragma unroll
      for (int u = 0; u < KU; u++) {
        const float4 kf = ld4(kh + 8 * u + 4 * half), qf = ld4(qh + 8 * u + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; i++) mean = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(kf, i), f4get(qf, i), mean, 0, 0, 0);
      }
    }
    const float f = scale / (float)H;
    const bool qok = n0 + c32 < N;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int key = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float val = mean[r] * f;
      total += (qok && key < M) ? val * val : 0.f;
    }
  }
  total = se3_wave_sum(total);
  if (lane == 0) red[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// grid (ceil(N/32), H, A): out[a, n, h] = sum_e mix[a, e] softmax_m(q_a.k_e * scale) v_e ; waves split the key anchors e
template <int D>
__global__ __launch_bounds__(256) void cross_eq_apply_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ vt, const float* __restrict__ mix,
                                                             int A, int N, int M, int C, int Mp, float scale, float* __restrict__ out) {
  constexpr int DT = FlashState<D>::DT;
  __shared__ float so[4 * DT * 16 * 64];
  const int n0 = blockIdx.x * 32, h = blockIdx.y, a = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  FlashState<D> tot;
  flash_init(tot);
  for (int e = wave; e < A; e += 4) {
    FlashState<D> st;
    flash_init(st);
    flash_tiles<D>(st, q + (size_t)a * N * C + h * D, k + (size_t)e * M * C + h * D, vt + ((size_t)e * C + h * D) * Mp, nullptr,
                   n0, N, M, C, C, Mp, Mp, scale, 0, 1);
    const float w = mix[a * A + e] / st.l;
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int r = 0; r < 16; r++) tot.o[dt][r] += st.o[dt][r] * w;
  }
#pragma unroll
  for (int dt = 0; dt < DT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) so[((wave * DT + dt) * 16 + r) * 64 + lane] = tot.o[dt][r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) acc += so[((w * DT + dt) * 16 + r) * 64 + lane];
        tot.o[dt][r] = acc;
      }
    flash_store<D>(tot, out + (size_t)a * N * C + h * D, n0, N, C, 1.f, false);
  }
}

// ---- stack mode of the equivariant cross attention: all pairs of a batch per launch ------------------------------------------
struct CrossEqArgs {
  const float *q, *k, *vt;     // q (A, Rq, C), k (A, Rk, C) packed rows; vt (A, C, v_rs) transposed values addressed by key column
  Stack S;                     // per pair: q_start, k_start, N, M (Mp = ceil32(M))
  int A, C, H, QT;             // QT = 32-query tiles of the largest pair
  long long q_sa, k_sa, v_sa;  // anchor strides (floats)
  int v_rs;
  float scale;
  int G;                       // cross_eq_apply_stack_x6_kernel: key-anchor groups (each writes its partial sum at channel offset g * C)
  int out_rs;                  //   row stride of its output (G * C) and
  long long out_sa;            //   anchor stride of its output
};

// grid (QT, A*A, pairs): partial[(pair*A*A + ae) * QT + qt] = sum over the tile's (n, m) of (mean_h S[a,e,h,n,m])^2
template <int D>
__global__ __launch_bounds__(256) void cross_eq_stats_stack_kernel(CrossEqArgs p, float* __restrict__ partial) {
  constexpr int KU = D / 8;
  __shared__ float red[4];
  const int A = p.A, C = p.C, H = p.H;
  const StackCloud cl = stack_pick(p.S, blockIdx.z);
  const int N = cl.N, M = cl.M;
  const int n0 = blockIdx.x * 32, ae = blockIdx.y, a = ae / A, e = ae - a * A;
  float* dst = partial + ((size_t)blockIdx.z * A * A + ae) * gridDim.x + blockIdx.x;
  if (n0 >= N) {
    if (threadIdx.x == 0) *dst = 0.f;
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, c32 = lane & 31;
  const int nq = min(n0 + c32, N - 1);
  const int tiles = (M + 31) >> 5;
  const float* qa = p.q + a * p.q_sa + (size_t)cl.q_start * C;
  const float* ke = p.k + e * p.k_sa + (size_t)cl.k_start * C;
  float total = 0.f;
  for (int tile = wave; tile < tiles; tile += 4) {
    const int m0 = tile << 5;
    const int kr = min(m0 + c32, M - 1);
    f32x16 mean;
#pragma unroll
    for (int r = 0; r < 16; r++) mean[r] = 0.f;
    for (int h = 0; h < H; h++) {
      const float* qh = qa + (size_t)nq * C + h * D;
      const float* kh = ke + (size_t)kr * C + h * D;
#pragma unroll
      for (int u = 0; u < KU; u++) {
        const float4 kf = ld4(kh + 8 * u + 4 * half), qf = ld4(qh + 8 * u + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; i++) mean = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(kf, i), f4get(qf, i), mean, 0, 0, 0);
      }
    }
    const float f = p.scale / (float)H;
    const bool qok = n0 + c32 < N;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int key = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float val = mean[r] * f;
      total += (qok && key < M) ? val * val : 0.f;
    }
  }
  total = se3_wave_sum(total);
  if (lane == 0) red[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) *dst = (red[0] + red[1]) + (red[2] + red[3]);
}

// grid (QT, H, A * pairs), 64 NW threads: wave e handles key anchor e (NW = A = 6: no imbalance, half the serial chain of the
// 4-wave version); out[a, n, h] = sum_e mix[pair, a, e] softmax_m(q_a.k_e * scale) v_e
template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void cross_eq_apply_stack_kernel(CrossEqArgs p, const float* __restrict__ mix,
                                                                       float* __restrict__ out) {
  constexpr int DT = FlashState<D>::DT;
  __shared__ float so[NW * DT * 16 * 64];
  const int A = p.A, C = p.C;
  const int pair = blockIdx.z / A, a = blockIdx.z - pair * A;
  const StackCloud cl = stack_pick(p.S, pair);
  const int n0 = blockIdx.x * 32, h = blockIdx.y;
  if (n0 >= cl.N) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  FlashState<D> tot;
  flash_init(tot);
  for (int e = wave; e < A; e += NW) {
    FlashState<D> st;
    flash_init(st);
    flash_tiles<D, false>(st, p.q + a * p.q_sa + (size_t)cl.q_start * C + h * D, p.k + e * p.k_sa + (size_t)cl.k_start * C + h * D,
                          p.vt + e * p.v_sa + (size_t)h * D * p.v_rs + cl.k_start, nullptr, n0, cl.N, cl.M, C, C, p.v_rs, cl.Mp,
                          p.scale, 0, 1);
    const float w = mix[((size_t)pair * A + a) * A + e] / st.l;
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int r = 0; r < 16; r++) tot.o[dt][r] += st.o[dt][r] * w;
  }
#pragma unroll
  for (int dt = 0; dt < DT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) so[((wave * DT + dt) * 16 + r) * 64 + lane] = tot.o[dt][r];
  __syncthreads();
  // every wave sums and stores its share of the output columns
  const int half = lane >> 5, c32 = lane & 31, nq = n0 + c32;
  float* op = out + ((size_t)a * p.q_sa) + (size_t)cl.q_start * C + h * D;
  for (int gi = wave; gi < DT * 4; gi += NW) {
    const int dt = gi >> 2, g = gi & 3;
    const int dd = 32 * dt + 8 * g + 4 * half;
    float val[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) acc += so[((w * DT + dt) * 16 + 4 * g + j) * 64 + lane];
      val[j] = acc;
    }
    if (nq < cl.N && dd < D) *reinterpret_cast<float4*>(op + (size_t)nq * C + dd) = make_float4(val[0], val[1], val[2], val[3]);
  }
}

// ---- split-f16 form of the equivariant cross attention (stack mode, head dimension 64) -------------------------------------------------
// cross_eq_apply_stack_kernel runs at the f32 MFMA rate (75 GF per call in 0.5 ms).  Here q, k and the transposed values are split once
// per call into f16 hi + lo pieces (x = hi + lo to 2^-22 |x|) and both products of the flash loop run on v_mfma_f32_32x32x16_f16 with the
// three piece products hi hi + hi lo + lo hi accumulated in f32 -- the scheme of csrc/kpconv_mfma.hip: 3 x 32 cycles per 16 k instead of
// 8 x 64 (round 2 used three bf16 pieces and six products: twice the matrix work and 1.5x the piece bytes for two more bits).  The softmax
// weights P are split on the fly from the S^T accumulator registers; their register order fixes the order of the keys inside an MFMA
// K-step (lane half h holds keys {0..3, 8..11} + 4 h of every 16), so the transposed values are stored with that permutation and a lane's
// V fragment is one 16-byte load.  (Entry point and kernel keep their round-2 names, "x6".)
typedef _Float16 h2x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void h2_split8(const float (&x)[8], uint4& hi, uint4& lo) {
  h2x8_t h, l;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    h[i] = (_Float16)x[i];
    l[i] = (_Float16)(x[i] - (float)h[i]);
  }
  hi = __builtin_bit_cast(uint4, h);
  lo = __builtin_bit_cast(uint4, l);
}

// The operand splits of the f16 attention kernels carry their own powers of two (round 5: the last part of the f16-window item, DESIGN
// section 4 "Range of the f16 split"): a query row of one head, a block of 8 key rows of one head and a value channel of one cloud are each
// brought to [2^6, 2^7) before the split -- exact, and the kernels take the scales out of the f32 logits (per lane: the query; per register
// group of the S^T tile: the key block) and of the output (per channel).  Nothing finite leaves the f16 range any more; a NaN / Inf in an
// operand becomes NaN pieces (the output rows it reaches are NaN, as with the reference's matmul + softmax) and the launch counts it
// (se3_debug_attention_saturated).
__device__ unsigned long long g_attn_saturated = 0;
__device__ __forceinline__ bool h2_split8_sat(const float (&x)[8], uint4& hi, uint4& lo) {
  float c[8];
  bool sat = false;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const bool nonfinite = (__float_as_uint(x[i]) & 0x7f800000u) == 0x7f800000u;
    c[i] = fminf(fmaxf(x[i], -65000.f), 65000.f);              // finite overflow: clamped and counted
    sat |= !(c[i] == x[i]);
    c[i] = nonfinite ? __builtin_bit_cast(float, 0x7fc00000) : c[i];      // NaN / Inf: NaN pieces -> NaN scores -> NaN rows, as the reference's matmul
  }
  h2_split8(c, hi, lo);
  return sat;
}
__device__ __forceinline__ void count_saturated(bool sat) {
  const unsigned long long m = __ballot(sat);
  if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&g_attn_saturated, (unsigned long long)__popcll(m));
}
// magnitude (integer maximum of |x| bit patterns: non-negative floats order like their bits) -> the power of two that brings it to
// [2^6, 2^7) and its inverse; zero or non-finite: 1; both stay normal floats (a denormal maximum gets 2^126)
struct SplitScale { float scale, inv; };
__device__ __forceinline__ SplitScale split_scale_of(unsigned amax_bits) {
  const int e = (int)(amax_bits >> 23) & 0xff;
  int k = amax_bits == 0u || e == 0xff ? 0 : 133 - e;          // (a NaN / Inf among the values: no scaling, the clamp takes it)
  k = k > 126 ? 126 : k;
  return SplitScale{__uint_as_float((unsigned)(127 + k) << 23), __uint_as_float((unsigned)(127 - k) << 23)};
}
__device__ __forceinline__ unsigned abs_bits_max8(const float (&v)[8]) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) m = max(m, __float_as_uint(v[i]) & 0x7fffffffu);
  return m;
}

// ONE launch splits every operand of a call: per anchor workgroups [0, q_groups) the query rows (cross attention only: the self-attention
// kernel splits its queries itself) and [q_groups, q_groups + k_groups) the key rows, then C / 4 workgroups for the transposed values;
// grid (A (q_groups + k_groups) + C / 4, clouds).
//   rows (A, R, row stride, anchor stride) -> pieces [2][A][R][C] f16; a wave = 8 rows x the 64 channels of one head, a lane 8 channels;
//     queries: scale per (row, head) -> qinv[(a H + h) R + row]; keys: per (8-row block, head) -> kinv[(a H + h) nblk + row / 8]
//     (key starts are multiples of 16: a block never spans two clouds; rows past the cloud's end do not count)
//   V^T (A, C, v_rs) -> pieces [2][A][C][v_rs] f16, the keys of every aligned block of 16 in the order 0..3, 8..11, 4..7, 12..15 (the
//     register order of P in the S^T accumulators); a wave = one channel of one cloud over all anchors (a lane: (anchor, 16-key block)
//     units, held in registers between the maximum and the split), scale per (cloud, channel) -> vinv[cloud C + c]: uniform along the
//     keys AND the anchors, so the kernels take it out once, in their epilogue
struct X6SplitArgs {
  Stack S;
  int A, C, H;
  const float *q, *k, *vt;
  int q_rs, k_rs, v_rs;
  long long q_sa, k_sa, v_sa;
  long long q_rows, k_rows;                 // rows of the piece tensors
  uint4 *outq, *outk, *outv;
  float *qinv, *kinv, *vinv;
  int q_groups, k_groups;                   // workgroups (4 waves) of the query / key part
};
__global__ __launch_bounds__(256) void x6_split_kernel(X6SplitArgs p) {
  const int ci = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const StackCloud cl = stack_pick(p.S, ci);
  bool sat = false;
  const int row_groups = p.q_groups + p.k_groups;
  if ((int)blockIdx.x < p.A * row_groups) {
    const int a = (int)blockIdx.x / row_groups, bx = (int)blockIdx.x - a * row_groups;
    const bool isq = bx < p.q_groups;
    const int item = (bx - (isq ? 0 : p.q_groups)) * 4 + wave;
    const int rg = item / p.H, h = item - rg * p.H;
    const int rows = isq ? cl.N : cl.M, start = isq ? cl.q_start : cl.k_start;
    if (rg * 8 < rows) {
      const int r = lane >> 3, c8 = lane & 7, row = rg * 8 + r;
      const bool live = row < rows;
      const float* x = isq ? p.q : p.k;
      const int rs = isq ? p.q_rs : p.k_rs;
      const long long sa = isq ? p.q_sa : p.k_sa, R = isq ? p.q_rows : p.k_rows;
      const float* src = x + a * sa + (int64_t)(start + (live ? row : rows - 1)) * rs + h * 64 + c8 * 8;
      const float4 lo = ld4(src), hi = ld4(src + 4);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      unsigned m = live ? abs_bits_max8(v) : 0u;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));          // the row's 64 channels
      if (!isq) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));       // the block's 8 rows
      }
      const SplitScale sc = split_scale_of(m);
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] *= sc.scale;
      uint4 p1, p2;
      const bool s1 = h2_split8_sat(v, p1, p2);
      if (live) {
        sat |= s1;
        uint4* out = isq ? p.outq : p.outk;
        const int64_t o = (((int64_t)a * R + start + row) * p.C + h * 64 + c8 * 8) >> 3, total = ((int64_t)p.A * R * p.C) >> 3;
        out[o] = p1;
        out[total + o] = p2;
        if (isq) {
          if (c8 == 0) p.qinv[((int64_t)a * p.H + h) * R + start + row] = sc.inv;
        } else if (lane == 0) {
          p.kinv[((int64_t)a * p.H + h) * ((R + 7) >> 3) + ((start + row) >> 3)] = sc.inv;
        }
      }
    }
  } else {
    const int c = ((int)blockIdx.x - p.A * row_groups) * 4 + wave;
    if (c < p.C) {
      const int nb = ((cl.M + 31) >> 5) << 1;                      // 16-key blocks of the cloud's padded key range
      const int items = p.A * nb;                                  // (anchor, block) units of this channel: lane l takes l, l + 64, ..
      const int64_t vblocks = p.v_rs >> 4, vtot = (int64_t)p.A * p.C * vblocks;
      const float* chan = p.vt + (int64_t)c * p.v_rs + cl.k_start;
      auto unit_max = [&](int b, const float4 (&x)[4]) {
        unsigned m = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int key = 16 * b + 4 * u;
          m = max(m, key + 0 < cl.M ? __float_as_uint(x[u].x) & 0x7fffffffu : 0u);
          m = max(m, key + 1 < cl.M ? __float_as_uint(x[u].y) & 0x7fffffffu : 0u);
          m = max(m, key + 2 < cl.M ? __float_as_uint(x[u].z) & 0x7fffffffu : 0u);
          m = max(m, key + 3 < cl.M ? __float_as_uint(x[u].w) & 0x7fffffffu : 0u);
        }
        return m;
      };
      auto emit = [&](int aa, int b, const float4 (&x)[4], float s) {
        // (the padding keys of the last tile are multiplied by P = 0: they only have to be finite -- whatever the caller's buffer holds
        //  there, they are stored as zeros and never counted)
        float v[16];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int key = 16 * b + 4 * u;
          v[4 * u + 0] = key + 0 < cl.M ? x[u].x * s : 0.f;
          v[4 * u + 1] = key + 1 < cl.M ? x[u].y * s : 0.f;
          v[4 * u + 2] = key + 2 < cl.M ? x[u].z * s : 0.f;
          v[4 * u + 3] = key + 3 < cl.M ? x[u].w * s : 0.f;
        }
        const float lo[8] = {v[0], v[1], v[2], v[3], v[8], v[9], v[10], v[11]};          // keys 0..3, 8..11
        const float hi[8] = {v[4], v[5], v[6], v[7], v[12], v[13], v[14], v[15]};        // keys 4..7, 12..15
        uint4 a1, a2, b1, b2;
        const bool s1 = h2_split8_sat(lo, a1, a2), s2 = h2_split8_sat(hi, b1, b2);
        sat |= s1 || s2;
        const int64_t j = ((int64_t)aa * p.C + c) * vblocks + (cl.k_start >> 4) + b;
        p.outv[2 * j] = a1; p.outv[2 * j + 1] = b1;
        p.outv[2 * (vtot + j)] = a2; p.outv[2 * (vtot + j) + 1] = b2;
      };
      constexpr int kHeld = 3;                                     // units a lane keeps in registers between the maximum and the split
      unsigned m = 0;
      if (items <= 64 * kHeld) {                                   // (6 anchors x 512 keys: every batch of the benches) one pass, all loads at once
        float4 x[kHeld][4];
        int ua[kHeld], ub[kHeld];
#pragma unroll
        for (int t = 0; t < kHeld; t++) {
          const int it = lane + 64 * t, ok = it < items;
          ua[t] = ok ? it / nb : 0;
          ub[t] = ok ? it - ua[t] * nb : 0;
#pragma unroll
          for (int u = 0; u < 4; u++) x[t][u] = ld4(chan + ua[t] * p.v_sa + 16 * ub[t] + 4 * u);
        }
#pragma unroll
        for (int t = 0; t < kHeld; t++) m = max(m, lane + 64 * t < items ? unit_max(ub[t], x[t]) : 0u);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        const SplitScale sc = split_scale_of(m);
#pragma unroll
        for (int t = 0; t < kHeld; t++)
          if (lane + 64 * t < items) emit(ua[t], ub[t], x[t], sc.scale);
        if (lane == 0) p.vinv[(int64_t)ci * p.C + c] = sc.inv;
      } else {                                                     // two passes, the second out of L1 / L2
        for (int it = lane; it < items; it += 64) {
          const int aa = it / nb, b = it - aa * nb;
          float4 x[4];
#pragma unroll
          for (int u = 0; u < 4; u++) x[u] = ld4(chan + aa * p.v_sa + 16 * b + 4 * u);
          m = max(m, unit_max(b, x));
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        const SplitScale sc = split_scale_of(m);
        for (int it = lane; it < items; it += 64) {
          const int aa = it / nb, b = it - aa * nb;
          float4 x[4];
#pragma unroll
          for (int u = 0; u < 4; u++) x[u] = ld4(chan + aa * p.v_sa + 16 * b + 4 * u);
          emit(aa, b, x, sc.scale);
        }
        if (lane == 0) p.vinv[(int64_t)ci * p.C + c] = sc.inv;
      }
    }
  }
  count_saturated(sat);
}
// sizes of the scale tables behind the pieces (floats); `rows` of keys -> ceil(rows / 8) blocks
static inline size_t x6_kinv_floats(int A, int C, int64_t k_rows) { return (size_t)A * (size_t)(C / 64 + 1) * (size_t)((k_rows + 7) / 8 + 4); }
static inline size_t x6_qinv_floats(int A, int C, int64_t q_rows) { return (size_t)A * (size_t)(C / 64 + 1) * (size_t)(q_rows + 4); }
static inline size_t x6_vinv_floats(int C) { return (size_t)kMaxClouds * (size_t)C; }

// Value of lane i ^ 32 (the other half of the wave): one v_permlane32_swap instead of a ds_bpermute round trip through the LDS pipeline.
__device__ __forceinline__ float other_half(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

struct X6Pieces {
  const uint4 *q[2], *k[2], *v[2];       // f16 hi / lo pieces, 8 values per uint4
  const float *qinv, *kinv, *vinv;       // inverse scales: [(a H + h) q_rows + row], [(e H + h) nblk + row / 8], [pair C + c] (x6_split_kernel)
  long long q_rows;
  int nblk;
};
// lane i ^ 32's value of an integer
__device__ __forceinline__ unsigned other_half_u32(unsigned v) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return (threadIdx.x & 32) ? r[0] : r[1];
}

// grid (ceil(QT / 4), H, A * pairs), 4 waves = 4 consecutive 32-query tiles of one (pair, query anchor a, head).  The workgroup walks
// the (key anchor e, 32-key tile) sequence once; every K / V^T tile (3 pieces each: 24 KB) goes global -> registers -> LDS one step ahead
// and is read by the four waves from LDS (rows padded to an odd number of 16-byte units: conflict-free ds_read_b128) -- with every wave
// fetching its own fragments from L1 (the first form of this kernel: one wave per key anchor) the 512 B per MFMA saturated L1's return
// path and the kernel ran at the f32 kernel's speed.  out[a, n, h] = sum_e mix[pair, a, e] softmax_m(q_a.k_e * scale) v_e.
constexpr int kX6KRow = 9, kX6VRow = 5;        // uint4 per K row (64 d = 8 used) / V^T row (32 keys = 4 used)
__global__ __launch_bounds__(256) void cross_eq_apply_stack_x6_kernel(CrossEqArgs p, X6Pieces X, int64_t q_piece_sa, int64_t k_piece_sa,
                                                                      int64_t v_piece_sa, const float* __restrict__ mix,
                                                                      float* __restrict__ out) {
  constexpr int D = 64;
  __shared__ uint4 ktile[2][2][32][kX6KRow];
  __shared__ uint4 vtile[2][2][64][kX6VRow];
  const int A = p.A, C = p.C;
  // G > 1 (few pairs: 72 workgroups per pair leave most of the chip idle while every wave walks A * tiles dependent steps): workgroup
  // (.., g) takes the key anchors [g, g + 1) * A / G and writes ITS weighted sum at channel offset g * C of a (A, Rq, G * C) output; the
  // output projection that follows adds the groups (stacked weights), no reduction pass and no atomics
  const int zz = blockIdx.z / p.G, grp = blockIdx.z - zz * p.G;
  const int pair = zz / A, a = zz - pair * A;
  const StackCloud cl = stack_pick(p.S, pair);
  const int h = blockIdx.y;
  if (blockIdx.x * 128 >= cl.N) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, c32 = lane & 31;
  const int n0 = blockIdx.x * 128 + wave * 32;
  const bool active = n0 < cl.N;                            // (inactive waves still help with the tile copies)
  const int nq = min(n0 + c32, cl.N - 1);
  const int64_t q_off = a * q_piece_sa + (int64_t)cl.q_start * C + h * D;
  h2x8_t qf[2][4];
#pragma unroll
  for (int pc = 0; pc < 2; pc++)
#pragma unroll
    for (int u = 0; u < 4; u++) qf[pc][u] = __builtin_bit_cast(h2x8_t, X.q[pc][(q_off + (int64_t)nq * C + 16 * u + 8 * half) >> 3]);
  // logits = (scaled q . scaled k) / (query scale x key-block scale) x 1 / sqrt(d): the lane's factor, the block's is multiplied in per step
  const float q_unscale = X.qinv[((int64_t)a * p.H + h) * X.q_rows + cl.q_start + nq] * p.scale;
  const int last_blk = (cl.M - 1) >> 3;
  const int tiles = (cl.M + 31) >> 5, steps = (A / p.G) * tiles, base = grp * steps;     // this group's steps: base .. base + steps
  // this thread's share of a tile copy: K: 2 pieces x 32 rows x 8 uint4 = 512 -> 2 per thread; V^T: 2 pieces x 64 rows x 4 uint4 = 512 -> 2
  // Three register sets: the tile of step s + 3 is requested at the start of step s and published (written to LDS) at the end of step
  // s + 2 -- two full steps for the L2 round trip.  With one set (distance 1) a step lasted as long as that round trip (~5 000 cycles for
  // 24 MFMAs + a 32-key softmax), the kernel was latency bound at 2.25 waves per SIMD.  Requests are unconditional (clamped) so that the
  // compiler counts them: an `if` around a load turns every later wait into vmcnt(0).
  u32x4r rk[3][2], rv[3][2];                               // (ext_vector_type: arrays of the HIP uint4 struct end up in scratch)
  float rs[3];                                             // lane l: the inverse scale of key block (l & 3) of the tile
  auto request = [&](int step, u32x4r (&rk)[2], u32x4r (&rv)[2], float& rs) {
    step = base + (step < steps ? step : steps - 1);
    const int e = step / tiles, m0 = (step - e * tiles) << 5;
    rs = X.kinv[((int64_t)e * p.H + h) * X.nblk + (cl.k_start >> 3) + min((m0 >> 3) + (lane & 3), last_blk)];
    const int64_t k_off = e * k_piece_sa + (int64_t)cl.k_start * C + h * D;
    const int64_t v_off = e * v_piece_sa + (int64_t)h * D * p.v_rs + cl.k_start + m0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = tid >> 3, q8 = tid & 7;               // piece i: 32 rows x 8 uint4
      rk[i] = __builtin_bit_cast(u32x4r, X.k[i][(k_off + (int64_t)min(m0 + row, cl.M - 1) * C + 8 * q8) >> 3]);
      const int vrow = tid >> 2, q4 = tid & 3;              // piece i: 64 rows x 4 uint4
      rv[i] = __builtin_bit_cast(u32x4r, X.v[i][(v_off + (int64_t)vrow * p.v_rs + 8 * q4) >> 3]);
    }
  };
  auto publish = [&](int buf, const u32x4r (&rk)[2], const u32x4r (&rv)[2]) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      ktile[buf][i][tid >> 3][tid & 7] = __builtin_bit_cast(uint4, rk[i]);
      vtile[buf][i][tid >> 2][tid & 3] = __builtin_bit_cast(uint4, rv[i]);
    }
  };
  request(0, rk[0], rv[0], rs[0]);
  publish(0, rk[0], rv[0]);
  request(1, rk[1], rv[1], rs[1]);
  request(2, rk[2], rv[2], rs[2]);
  __syncthreads();
  FlashState<D> tot, st;
  flash_init(tot);
  flash_init(st);
  // one step; RS: the register set that is free now (tile step + 3 goes there), PS: the set holding tile step + 1
  auto one_step = [&](int step, u32x4r (&rk_req)[2], u32x4r (&rv_req)[2], float& rs_req, const u32x4r (&rk_pub)[2],
                      const u32x4r (&rv_pub)[2]) {
    const int buf = step & 1;
    const int e = (base + step) / tiles, tile = base + step - e * tiles, m0 = tile << 5;
    const float ks = rs_req;                                  // (this step's block scales sit in the set about to be refilled)
    request(step + 3, rk_req, rv_req, rs_req);
    if (active && step < steps) {
      // two accumulators, consecutive MFMAs alternate between them
      f32x16 s, s2;
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] = s2[r] = 0.f;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        h2x8_t kf[2];
#pragma unroll
        for (int pc = 0; pc < 2; pc++) kf[pc] = __builtin_bit_cast(h2x8_t, ktile[buf][pc][c32][2 * u + half]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[0][u], s, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[1][u], s2, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0][u], s, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] += s2[r];
      const float unscale[4] = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks), 0)) * q_unscale,
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks), 1)) * q_unscale,
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks), 2)) * q_unscale,
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks), 3)) * q_unscale};
      float mx = -INFINITY;
#pragma unroll
      for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int key = m0 + 8 * g + 4 * half + j;
          float val = s[4 * g + j] * unscale[g];
          val = key < cl.M ? val : -INFINITY;
          s[4 * g + j] = val;
          mx = fmaxf(mx, val);
        }
      }
      mx = fmaxf(mx, other_half(mx));
      const float m_new = fmaxf(st.m, mx);
      const float alpha = __expf(st.m - m_new);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        s[r] = __expf(s[r] - m_new);
        ps += s[r];
      }
      ps += other_half(ps);
      st.l = st.l * alpha + ps;
      st.m = m_new;
      h2x8_t pb[2][2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const float v8[8] = {s[8 * j], s[8 * j + 1], s[8 * j + 2], s[8 * j + 3], s[8 * j + 4], s[8 * j + 5], s[8 * j + 6], s[8 * j + 7]};
        uint4 p1, p2;
        h2_split8(v8, p1, p2);
        pb[0][j] = __builtin_bit_cast(h2x8_t, p1);
        pb[1][j] = __builtin_bit_cast(h2x8_t, p2);
      }
#pragma unroll
      for (int dt = 0; dt < 2; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) st.o[dt][r] *= alpha;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        h2x8_t vf[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
          for (int pc = 0; pc < 2; pc++) vf[dt][pc] = __builtin_bit_cast(h2x8_t, vtile[buf][pc][32 * dt + c32][2 * j + half]);
#define SE3_X6_PV(a_, b_)                                                                                                       \
  st.o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0][a_], pb[b_][j], st.o[0], 0, 0, 0);                                     \
  st.o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1][a_], pb[b_][j], st.o[1], 0, 0, 0);
        SE3_X6_PV(1, 0) SE3_X6_PV(0, 1) SE3_X6_PV(0, 0)
#undef SE3_X6_PV
      }
      if (tile == tiles - 1) {                                // key anchor e is complete: fold it into the output with its weight
        const float w = mix[((size_t)pair * A + a) * A + e] / st.l;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
          for (int r = 0; r < 16; r++) tot.o[dt][r] += st.o[dt][r] * w;
        flash_init(st);
      }
    }
    publish(buf ^ 1, rk_pub, rv_pub);                          // its readers finished before the barrier that ended the previous step
    __syncthreads();
  };
  // (unconditional three-step body: see attention_x6_kernel; steps = A * tiles, padded steps fall on key anchor e >= A and are skipped whole)
  for (int step = 0; step < steps; step += 3) {                // set of tile t = t % 3 (tile 0 went through set 0 above)
    one_step(step, rk[0], rv[0], rs[0], rk[1], rv[1]);
    one_step(step + 1, rk[1], rv[1], rs[1], rk[2], rv[2]);
    one_step(step + 2, rk[2], rv[2], rs[2], rk[0], rv[0]);
  }
  if (!active) return;
  // o[dt][r] = O^T[d = 32 dt + (r & 3) + 8 (r >> 2) + 4 half][query c32]: four consecutive d per (dt, g) -> one float4 per lane; the
  // values' channel scales (uniform over keys and key anchors) leave here
  const int nrow = n0 + c32;
  if (nrow < cl.N) {
    float* op = out + ((size_t)a * p.out_sa) + ((size_t)cl.q_start + nrow) * p.out_rs + grp * C + h * D;
    const float* vi = X.vinv + (size_t)pair * C + h * D;
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const float4 w = ld4(vi + 32 * dt + 8 * g + 4 * half);
        *reinterpret_cast<float4*>(op + 32 * dt + 8 * g + 4 * half) =
            make_float4(tot.o[dt][4 * g] * w.x, tot.o[dt][4 * g + 1] * w.y, tot.o[dt][4 * g + 2] * w.z, tot.o[dt][4 * g + 3] * w.w);
      }
  }
}

// one workgroup: g[a,e] = sum_i partial[(a*A+e)*P + i] / (N M) -> mixing weights.
//   mode 0 (a_soft): mix[a,e] = g[a,e] / sum_e g[a,e];  weights = mix (A*A values)
//   mode 1 (r_soft): w[r] = mean_a g[a, trace[r,a]] normalised over r; mix[a,e] = sum_{r: trace[r,a]=e} w[r]; weights = w (R values)
// grid = pairs: block b works on partial + b * A*A*P, mix + b * A*A, weights + b * (mode 0: A*A, mode 1: R); inv_nm: 1 / (N M)
// of the pair, taken from the stack descriptor when S != nullptr
__global__ __launch_bounds__(64) void cross_eq_mix_kernel(const float* __restrict__ partial, int P, float inv_nm, int A, int R,
                                                          const int64_t* __restrict__ trace, int mode, float* __restrict__ mix,
                                                          float* __restrict__ weights, const Stack* __restrict__ S_dev, Stack S,
                                                          int use_stack) {
  __shared__ float g[64], w[64], tot;
  const int t = threadIdx.x;
  (void)S_dev;
  if (use_stack) {
    const StackCloud cl = stack_pick(S, blockIdx.x);
    inv_nm = 1.0f / ((float)cl.N * (float)cl.M);
  }
  partial += (size_t)blockIdx.x * A * A * P;
  mix += (size_t)blockIdx.x * A * A;
  weights += (size_t)blockIdx.x * (mode == 0 ? A * A : R);
  if (t < A * A) {
    float acc = 0.f;
    for (int i = 0; i < P; i++) acc += partial[(size_t)t * P + i];
    g[t] = acc * inv_nm;
  }
  __syncthreads();
  if (mode == 0) {
    if (t < A * A) {
      const int a = t / A;
      float rs = 0.f;
      for (int e = 0; e < A; e++) rs += g[a * A + e];
      const float v = g[t] / rs;
      mix[t] = v;
      weights[t] = v;
    }
    return;
  }
  if (t < R) {
    float acc = 0.f;
    for (int a = 0; a < A; a++) acc += g[a * A + (int)trace[t * A + a]];
    w[t] = acc / (float)A;
  }
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int r = 0; r < R; r++) s += w[r];
    tot = s;
  }
  __syncthreads();
  if (t < R) {
    w[t] = w[t] / tot;
    weights[t] = w[t];
  }
  __syncthreads();
  if (t < A * A) {
    const int a = t / A, e = t - a * A;
    float acc = 0.f;
    for (int r = 0; r < R; r++) acc += ((int)trace[r * A + a] == e) ? w[r] : 0.f;
    mix[t] = acc;
  }
}

template <typename F>
int dispatch_head_dim(int D, F&& f, const char* what) {
  switch (D) {
    case 8: f(std::integral_constant<int, 8>()); return SE3_OK;
    case 16: f(std::integral_constant<int, 16>()); return SE3_OK;
    case 32: f(std::integral_constant<int, 32>()); return SE3_OK;
    case 64: f(std::integral_constant<int, 64>()); return SE3_OK;
    default:
      se3_set_error("%s: head dim %d not in {8, 16, 32, 64}", what, D);
      return SE3_ERR_UNSUPPORTED;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// per-launch kernel timing (bench.py): with timing enabled the two RPE self-attention kernels are launched through
// hipExtLaunchKernelGGL with a start / stop HIP event pair each, recorded on the launch stream; the events carry the dispatch's
// own begin / end timestamps (what rocprofv3 --kernel-trace reports), without the marker-packet and dispatch-gap overhead of
// hipEventRecord brackets.
// ---------------------------------------------------------------------------------------------------------------------
struct TimedLaunch {
  hipEvent_t start, stop;
  int tag;              // 1 = rpe_bias_kernel, 2 = attention_kernel
  double aux;           // tag 1 inside se3_rpe_self_attention_stack*_fwd: the call's algorithmic bytes, negative for an equivariant call
};
static thread_local double g_next_aux = 0.0;      // set by the stack-mode self-attention entry for its logits launch
static bool g_time_kernels = false;
static std::vector<TimedLaunch> g_timed;
static std::recursive_mutex g_timed_mutex;      // several host threads may launch (one HIP stream each)

template <typename K, typename... Args>
static void launch_kernel(int tag, K kernel, dim3 grid, dim3 block, hipStream_t st, Args... args) {
  if (g_time_kernels) {
    TimedLaunch t{nullptr, nullptr, tag, tag == 1 ? g_next_aux : 0.0};
    if (tag == 1) g_next_aux = 0.0;
    if (hipEventCreate(&t.start) == hipSuccess && hipEventCreate(&t.stop) == hipSuccess) {
      std::lock_guard<std::recursive_mutex> lock(g_timed_mutex);
      hipExtLaunchKernelGGL(kernel, grid, block, 0, st, t.start, t.stop, 0, args...);
      g_timed.push_back(t);
      return;
    }
  }
  hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
}

extern "C" void se3_debug_kernel_timing(int enable) { g_time_kernels = enable != 0; }

// Waits for the recorded launches, writes their durations (microseconds) and tags in launch order, releases the events and
// returns the number of launches recorded (entries beyond `capacity` are dropped).
extern "C" int se3_debug_kernel_timing_collect(float* microseconds, int* tags, int capacity) {
  std::lock_guard<std::recursive_mutex> lock(g_timed_mutex);
  int n = 0;
  for (const TimedLaunch& t : g_timed) {
    float ms = 0.f;
    if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess && n < capacity) {
      microseconds[n] = ms * 1e3f;
      tags[n] = t.tag;
      n++;
    }
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  g_timed.clear();
  return n;
}

// The same with the algorithmic bytes recorded for the logits launches of the stack-mode self-attention calls (0 elsewhere): the caller
// needs no record of its own of which call was which (the launches may come from se3_transformer_forward).
extern "C" int se3_debug_kernel_timing_collect_ex(float* microseconds, int* tags, double* aux, int capacity) {
  std::lock_guard<std::recursive_mutex> lock(g_timed_mutex);
  int n = 0;
  for (const TimedLaunch& t : g_timed) {
    float ms = 0.f;
    if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess && n < capacity) {
      microseconds[n] = ms * 1e3f;
      tags[n] = t.tag;
      aux[n] = t.aux;
      n++;
    }
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  g_timed.clear();
  return n;
}

static int g_attn_variant = 0;
static long long* g_attn_prof = nullptr;
extern "C" void se3_debug_set_attention_variant(int variant) { g_attn_variant = variant; }
extern "C" void se3_debug_set_attention_profile(long long* stamps) { g_attn_prof = stamps; }
static int g_bias_variant = 0;
static int g_bias_split = 0;
// tuning hooks (benchmarks only): kernel variant / m-split override; 0 = default
extern "C" void se3_debug_set_bias_variant(int variant, int split) { g_bias_variant = variant; g_bias_split = split; }

static int device_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

static int launch_rpe_bias(const float* qp, const float* qe, int row_stride, int64_t anchor_stride, Stack& S, int C, int AH,
                           int H, float* bias, hipStream_t st, bool emb_bf16 = false) {
  long long total = 0;
  for (int c = 0; c < S.n; c++) {
    const StackCloud& cl = S.c[c];
    // 32-bit lane offsets inside a cloud's embedding row block / equivariant embedding / logits block
    SE3_REQUIRE((long long)cl.M * C < (1ll << 31) && (long long)(AH / H) * cl.N * cl.M * 4 < (1ll << 31) &&
                    (long long)AH * cl.N * cl.Mp < (1ll << 31),
                SE3_ERR_UNSUPPORTED, "rpe_bias: cloud %d too large for 32-bit offsets (N %d, M %d)", c, cl.N, cl.M);
    S.c[c].unit_begin = (int)total;
    total += (long long)cl.N * (cl.Mp / 32);
  }
  SE3_REQUIRE(total < (1ll << 31), SE3_ERR_UNSUPPORTED, "rpe_bias: too many (row, key tile) units");
  S.total_units = (int)total;
  if (g_bias_variant == 9)      // diagnostic (tools/micro/rpe_eq_breakdown.py): every cloud writes its logits over the first cloud's block -- 1/16 of the bytes
    for (int c = 1; c < S.n; c++) S.c[c].bias_off = S.c[0].bias_off;
  // one resident round, each workgroup with a balanced range of units: 3 workgroups of 4 waves per CU (LDS: fragments + staging) for the
  // f32 / bf16-embedding kernels, 2 for the f16-split kernel (the splits need ~190 registers; 8 waves x 16 KB stay in flight per CU)
  const bool half_split = !emb_bf16 && g_bias_variant != 2 && g_bias_variant != 3 && (qe == nullptr || H % 4 == 0);
  // (round 4: with the quad-contiguous requests the f16-split kernel of <= 16 folded-query rows takes 130 registers: three resident
  // workgroups per compute unit, 337 against 351 us per invariant call; the 24-row form takes 176: two)
  int wgs = (half_split && !(AH <= 16 && g_bias_variant != 5) ? 2 : 3) * device_cu_count();
  if (g_bias_split > 0) wgs = g_bias_split * device_cu_count();
  if (wgs > (total + 3) / 4) wgs = (int)((total + 3) / 4);
  if (wgs < 1) wgs = 1;
  dim3 grid((unsigned)wgs);
#define SE3_BIAS_ARGS qp, qe, row_stride, anchor_stride, S, AH, H, bias
#define SE3_BIAS_LAUNCH_RT(CT, RT)                                                                                   \
  if (emb_bf16) {                                                                                                    \
    if (qe == nullptr) launch_kernel(1, rpe_bias_kernel<CT, RT, 3, false, true, true>, grid, dim3(256), st, SE3_BIAS_ARGS);           \
    else if (H % 4 == 0) launch_kernel(1, rpe_bias_kernel<CT, RT, 3, true, true, true>, grid, dim3(256), st, SE3_BIAS_ARGS);          \
    else launch_kernel(1, rpe_bias_kernel<CT, RT, 2, true, false, true>, grid, dim3(256), st, SE3_BIAS_ARGS);                         \
  } else if (qe == nullptr) {                                                                                               \
    if (g_bias_variant == 2) launch_kernel(1, rpe_bias_kernel<CT, RT, 2, false, true>, grid, dim3(256), st, SE3_BIAS_ARGS);           \
    else if (g_bias_variant == 3) launch_kernel(1, rpe_bias_kernel<CT, RT, 3, false, true>, grid, dim3(256), st, SE3_BIAS_ARGS);      \
    else if (g_bias_variant == 5) launch_kernel(1, rpe_bias_kernel<CT, RT, 2, false, true, false, true>, grid, dim3(256), st, SE3_BIAS_ARGS);       \
    else launch_kernel(1, rpe_bias_kernel<CT, RT, 2, false, true, false, true, true>, grid, dim3(256), st, SE3_BIAS_ARGS);            \
  } else if (H % 4 == 0) {                                                                                           \
    if (g_bias_variant == 2) launch_kernel(1, rpe_bias_kernel<CT, RT, 2, true, true>, grid, dim3(256), st, SE3_BIAS_ARGS);            \
    else if (g_bias_variant == 3) launch_kernel(1, rpe_bias_kernel<CT, RT, 3, true, true>, grid, dim3(256), st, SE3_BIAS_ARGS);       \
    else if (g_bias_variant == 5) launch_kernel(1, rpe_bias_kernel<CT, RT, 2, true, true, false, true>, grid, dim3(256), st, SE3_BIAS_ARGS);        \
    else launch_kernel(1, rpe_bias_kernel<CT, RT, 2, true, true, false, true, true>, grid, dim3(256), st, SE3_BIAS_ARGS);             \
  } else {                                                                                                           \
    launch_kernel(1, rpe_bias_kernel<CT, RT, 2, true, false>, grid, dim3(256), st, SE3_BIAS_ARGS);                                    \
  }
#define SE3_BIAS_LAUNCH(CT)                                                                                          \
  if (AH <= 16) {                                                                                                    \
    SE3_BIAS_LAUNCH_RT(CT, 1)                                                                                        \
  } else {                                                                                                           \
    SE3_BIAS_LAUNCH_RT(CT, 2)                                                                                        \
  }
  switch (C) {
    case 32: SE3_BIAS_LAUNCH(2) break;
    case 64: SE3_BIAS_LAUNCH(4) break;
    case 128: SE3_BIAS_LAUNCH(8) break;
    case 256: SE3_BIAS_LAUNCH(16) break;
    default:
      se3_set_error("rpe_bias: channels %d not in {32, 64, 128, 256}", C);
      return SE3_ERR_UNSUPPORTED;
  }
#undef SE3_BIAS_LAUNCH
#undef SE3_BIAS_LAUNCH_RT
#undef SE3_BIAS_ARGS
  SE3_CHECK_LAUNCH("rpe_bias");
  return SE3_OK;
}

extern "C" int se3_rpe_bias_fwd(const float* qp, const float* qe, int row_stride, int64_t anchor_stride, const float* emb,
                                const float* eq_emb, int N, int M, int C, int AH, int H, int bias_row_stride, float* bias,
                                void* stream) {
  SE3_REQUIRE(qp && emb && bias, SE3_ERR_INVALID_ARG, "rpe_bias: null pointer");
  SE3_REQUIRE(row_stride % 4 == 0 && row_stride >= H * C, SE3_ERR_INVALID_ARG, "rpe_bias: folded-query row stride");
  SE3_REQUIRE((qe == nullptr) == (eq_emb == nullptr), SE3_ERR_INVALID_ARG, "rpe_bias: qe and eq_emb go together");
  SE3_REQUIRE(N >= 1 && M >= 1 && AH >= 1 && AH <= 32 && H >= 1 && AH % H == 0, SE3_ERR_UNSUPPORTED,
              "rpe_bias: N %d M %d AH %d H %d", N, M, AH, H);
  SE3_REQUIRE(bias_row_stride >= M && bias_row_stride % 32 == 0, SE3_ERR_INVALID_ARG,
              "rpe_bias: the logits row stride must be a multiple of 32 covering M (all of it is written, zeros beyond M)");
  Stack S{};
  S.n = 1;
  S.c[0] = StackCloud{emb, eq_emb, 0, 0, N, M, bias_row_stride, 0, 0};
  return launch_rpe_bias(qp, qe, row_stride, anchor_stride, S, C, AH, H, bias, (hipStream_t)stream);
}

static int rpe_bias_stack(const float* qp, const float* qe, int row_stride, int64_t anchor_stride, const void* const* emb_ptrs,
                          const float* const* eq_ptrs, const int64_t* q_starts, const int64_t* q_lengths,
                          const int64_t* k_lengths, const int64_t* bias_offsets, int num_clouds, int C, int AH, int H,
                          float* bias, void* stream, bool emb_bf16) {
  SE3_REQUIRE(qp && emb_ptrs && q_starts && q_lengths && k_lengths && bias_offsets && bias, SE3_ERR_INVALID_ARG,
              "rpe_bias_stack: null pointer");
  SE3_REQUIRE(num_clouds >= 1 && num_clouds <= kMaxClouds, SE3_ERR_UNSUPPORTED, "rpe_bias_stack: %d clouds (1..%d)", num_clouds,
              kMaxClouds);
  SE3_REQUIRE(row_stride % 4 == 0 && row_stride >= H * C, SE3_ERR_INVALID_ARG, "rpe_bias_stack: folded-query row stride");
  SE3_REQUIRE(AH >= 1 && AH <= 32 && H >= 1 && AH % H == 0, SE3_ERR_UNSUPPORTED, "rpe_bias_stack: AH %d H %d", AH, H);
  Stack S{};
  S.n = num_clouds;
  for (int c = 0; c < num_clouds; c++) {
    SE3_REQUIRE(emb_ptrs[c] != nullptr && q_lengths[c] >= 1 && k_lengths[c] >= 1 && q_starts[c] >= 0 && bias_offsets[c] >= 0 &&
                    bias_offsets[c] % 4 == 0,
                SE3_ERR_INVALID_ARG, "rpe_bias_stack: cloud %d descriptor", c);
    SE3_REQUIRE((qe == nullptr) == (eq_ptrs == nullptr || eq_ptrs[c] == nullptr), SE3_ERR_INVALID_ARG,
                "rpe_bias_stack: qe and the equivariant embeddings go together");
    const int M = (int)k_lengths[c];
    SE3_REQUIRE(((uintptr_t)emb_ptrs[c] & 15) == 0, SE3_ERR_INVALID_ARG, "rpe_bias_stack: embedding %d is not 16-byte aligned", c);
    S.c[c] = StackCloud{static_cast<const float*>(emb_ptrs[c]), qe ? eq_ptrs[c] : nullptr, (int)q_starts[c], 0, (int)q_lengths[c], M, ((M + 31) / 32) * 32,
                        0, (long long)bias_offsets[c]};
  }
  return launch_rpe_bias(qp, qe, row_stride, anchor_stride, S, C, AH, H, bias, (hipStream_t)stream, emb_bf16);
}

extern "C" int se3_rpe_bias_stack_fwd(const float* qp, const float* qe, int row_stride, int64_t anchor_stride,
                                      const float* const* emb_ptrs, const float* const* eq_ptrs, const int64_t* q_starts,
                                      const int64_t* q_lengths, const int64_t* k_lengths, const int64_t* bias_offsets,
                                      int num_clouds, int C, int AH, int H, float* bias, void* stream) {
  return rpe_bias_stack(qp, qe, row_stride, anchor_stride, reinterpret_cast<const void* const*>(emb_ptrs), eq_ptrs, q_starts,
                        q_lengths, k_lengths, bias_offsets, num_clouds, C, AH, H, bias, stream, false);
}

extern "C" int se3_rpe_bias_stack_bf16_fwd(const float* qp, const float* qe, int row_stride, int64_t anchor_stride,
                                           const uint16_t* const* emb_ptrs, const float* const* eq_ptrs,
                                           const int64_t* q_starts, const int64_t* q_lengths, const int64_t* k_lengths,
                                           const int64_t* bias_offsets, int num_clouds, int C, int AH, int H, float* bias,
                                           void* stream) {
  SE3_REQUIRE(C % 32 == 0, SE3_ERR_UNSUPPORTED, "rpe_bias_stack_bf16: channels %d not a multiple of 32", C);
  return rpe_bias_stack(qp, qe, row_stride, anchor_stride, reinterpret_cast<const void* const*>(emb_ptrs), eq_ptrs, q_starts,
                        q_lengths, k_lengths, bias_offsets, num_clouds, C, AH, H, bias, stream, true);
}

// ---- RPE / plain attention of a stack of clouds on the f16 matrix cores (head dimension 64) ---------------------------------------------
// attention_kernel above multiplies on the f32 matrix cores (57 % of their peak at A = 6: 160 us per call) and its f16-split form
// (MODE 4) splits every K and V^T tile again in every workgroup that reads it -- vector-ALU bound, no gain.  Here K and V^T are split ONCE
// per call (x6_split_kernel: the pieces have the byte size of the f32 tensors) and the kernel has the structure of
// cross_eq_apply_stack_x6_kernel: 4 waves = 4 consecutive 32-query tiles of one (cloud, anchor, head), every 32-key K / V^T tile goes
// global -> registers -> LDS THREE steps ahead and is read by the four waves from LDS, the logits tile of a wave (16 floats per lane, written
// by rpe_bias_kernel on other XCDs: a trip to the memory side) is requested three steps ahead as well.  No merge over waves: a wave owns
// its queries.  S = (q.k [+ bias]) * scale, online softmax, P split into f16 hi / lo in registers, three products each for q.k and P.v.
struct AttnPieces {
  const uint4 *k[2], *v[2];
  const float *kinv, *vinv;       // inverse scales: [(a H + h) nblk + row / 8], [cloud C + c] (x6_split_kernel)
  int nblk;
};

template <bool HAS_BIAS, bool PROF = false>
__global__ __launch_bounds__(256, HAS_BIAS ? 2 : 3) void attention_x6_kernel(AttnArgs p, AttnPieces X, int64_t k_piece_sa, int64_t v_piece_sa) {
  constexpr int D = 64;
  long long* prof = nullptr;                                // profiling hook (attention variant 12): 32 clock64() stamps per wave
  __shared__ uint4 ktile[2][2][32][kX6KRow];
  __shared__ uint4 vtile[2][2][64][kX6VRow];
  const int A = p.A, C = p.C;
  const int ci = blockIdx.z / A, a = blockIdx.z - ci * A;
  const StackCloud cl = stack_pick(p.S, ci);
  const int h = blockIdx.y;
  if (blockIdx.x * 128 >= cl.N) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, c32 = lane & 31;
  const int n0 = blockIdx.x * 128 + wave * 32;
  const bool active = n0 < cl.N;                            // (inactive waves still help with the tile copies)
  if (PROF && p.prof != nullptr) prof = p.prof + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 32;
  SE3_STAMP(0)
  const int nq = min(n0 + c32, cl.N - 1);
  h2x8_t qf[2][4];
  float q_unscale;              // logits = ((scaled q . scaled k) / (query scale x key-block scale) + bias) / sqrt(d): the lane's factor
  {
    const float* qr = p.q + a * p.q_sa + ((size_t)cl.q_start + nq) * p.q_rs + h * D + 8 * half;
    float v8[4][8];
    unsigned m = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float4 lo = ld4(qr + 16 * u), hi = ld4(qr + 16 * u + 4);
      v8[u][0] = lo.x; v8[u][1] = lo.y; v8[u][2] = lo.z; v8[u][3] = lo.w;
      v8[u][4] = hi.x; v8[u][5] = hi.y; v8[u][6] = hi.z; v8[u][7] = hi.w;
      m = max(m, abs_bits_max8(v8[u]));
    }
    m = max(m, other_half_u32(m));                          // the query's 64 channels of this head: its own power of two
    const SplitScale sc = split_scale_of(m);
    q_unscale = sc.inv * p.scale;
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) v8[u][i] *= sc.scale;         // (a NaN / Inf query stays one: its own output row, nobody else's)
      uint4 p1, p2;
      h2_split8(v8[u], p1, p2);
      qf[0][u] = __builtin_bit_cast(h2x8_t, p1);
      qf[1][u] = __builtin_bit_cast(h2x8_t, p2);
    }
  }
  const int last_blk = (cl.M - 1) >> 3;
  const float* kinv_row = X.kinv + ((int64_t)a * p.H + h) * X.nblk + (cl.k_start >> 3);
  const int steps = (cl.M + 31) >> 5;
  const int64_t k_off = a * k_piece_sa + (int64_t)cl.k_start * C + h * D;
  const int64_t v_off0 = a * v_piece_sa + (int64_t)h * D * p.v_rs + cl.k_start;
  const float* bias_row = HAS_BIAS ? p.bias + cl.bias_off + ((size_t)(a * p.H + h) * cl.N + nq) * cl.Mp + 4 * half : nullptr;
  u32x4r rk[3][2], rv[3][2];
  f32x4 rb[3][4];
  float rs[3];                                             // lane l: the inverse scale of key block (l & 3) of the tile
  auto request = [&](int step, u32x4r (&rk)[2], u32x4r (&rv)[2], f32x4 (&rb)[4], float& rs) {
    step = step < steps ? step : steps - 1;
    const int m0 = step << 5;
    rs = kinv_row[min((m0 >> 3) + (lane & 3), last_blk)];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = tid >> 3, q8 = tid & 7;               // piece i: 32 rows x 8 uint4
      rk[i] = __builtin_bit_cast(u32x4r, X.k[i][(k_off + (int64_t)min(m0 + row, cl.M - 1) * C + 8 * q8) >> 3]);
      const int vrow = tid >> 2, q4 = tid & 3;              // piece i: 64 rows x 4 uint4
      rv[i] = __builtin_bit_cast(u32x4r, X.v[i][(v_off0 + m0 + (int64_t)vrow * p.v_rs + 8 * q4) >> 3]);
    }
    if (HAS_BIAS) {
#pragma unroll
      for (int g = 0; g < 4; g++) rb[g] = *reinterpret_cast<const f32x4*>(bias_row + m0 + 8 * g);
    }
  };
  auto publish = [&](int buf, const u32x4r (&rk)[2], const u32x4r (&rv)[2]) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      ktile[buf][i][tid >> 3][tid & 7] = __builtin_bit_cast(uint4, rk[i]);
      vtile[buf][i][tid >> 2][tid & 3] = __builtin_bit_cast(uint4, rv[i]);
    }
  };
  request(0, rk[0], rv[0], rb[0], rs[0]);
  publish(0, rk[0], rv[0]);
  request(1, rk[1], rv[1], rb[1], rs[1]);
  request(2, rk[2], rv[2], rb[2], rs[2]);
  __syncthreads();
  SE3_STAMP(1)
  FlashState<D> st;
  flash_init(st);
  // one step; *_req: the register set that is free now (tile step + 3 goes there), *_pub: the set holding tile step + 1; bias: this step's
  auto one_step = [&](int step, u32x4r (&rk_req)[2], u32x4r (&rv_req)[2], f32x4 (&rb_req)[4], float& rs_req, const u32x4r (&rk_pub)[2],
                      const u32x4r (&rv_pub)[2]) {
    const int buf = step & 1, m0 = step << 5;
    f32x4 b4[4];
#pragma unroll
    for (int g = 0; g < 4; g++) b4[g] = HAS_BIAS ? rb_req[g] : f32x4{0.f, 0.f, 0.f, 0.f};   // (this step's logits sit in the set about to be refilled)
    const float ks = rs_req;                                                                //  and so do its key-block scales
    request(step + 3, rk_req, rv_req, rb_req, rs_req);
    if (PROF && step < 5) { SE3_STAMP(2 + 5 * step) }
    if (active) {
      f32x16 s, s2;
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] = s2[r] = 0.f;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        h2x8_t kf[2];
#pragma unroll
        for (int pc = 0; pc < 2; pc++) kf[pc] = __builtin_bit_cast(h2x8_t, ktile[buf][pc][c32][2 * u + half]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[0][u], s, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[1][u], s2, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0][u], s, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] += s2[r];
      const float unscale[4] = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks), 0)) * q_unscale,
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks), 1)) * q_unscale,
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks), 2)) * q_unscale,
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks), 3)) * q_unscale};
      float mx = -INFINITY;
#pragma unroll
      for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int key = m0 + 8 * g + 4 * half + j;
          float val = fmaf(s[4 * g + j], unscale[g], b4[g][j] * p.scale);
          val = key < cl.M ? val : -INFINITY;
          s[4 * g + j] = val;
          mx = fmaxf(mx, val);
        }
      }
      if (PROF && step < 5) { SE3_STAMP(3 + 5 * step) }
      mx = fmaxf(mx, other_half(mx));
      const float m_new = fmaxf(st.m, mx);
      const float alpha = __expf(st.m - m_new);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        s[r] = __expf(s[r] - m_new);
        ps += s[r];
      }
      ps += other_half(ps);
      st.l = st.l * alpha + ps;
      st.m = m_new;
      h2x8_t pb[2][2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const float v8[8] = {s[8 * j], s[8 * j + 1], s[8 * j + 2], s[8 * j + 3], s[8 * j + 4], s[8 * j + 5], s[8 * j + 6], s[8 * j + 7]};
        uint4 p1, p2;
        h2_split8(v8, p1, p2);
        pb[0][j] = __builtin_bit_cast(h2x8_t, p1);
        pb[1][j] = __builtin_bit_cast(h2x8_t, p2);
      }
      if (PROF && step < 5) { SE3_STAMP(4 + 5 * step) }
#pragma unroll
      for (int dt = 0; dt < 2; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) st.o[dt][r] *= alpha;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        h2x8_t vf[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
          for (int pc = 0; pc < 2; pc++) vf[dt][pc] = __builtin_bit_cast(h2x8_t, vtile[buf][pc][32 * dt + c32][2 * j + half]);
#define SE3_X6_PV(a_, b_)                                                                                                       \
  st.o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0][a_], pb[b_][j], st.o[0], 0, 0, 0);                                     \
  st.o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1][a_], pb[b_][j], st.o[1], 0, 0, 0);
        SE3_X6_PV(1, 0) SE3_X6_PV(0, 1) SE3_X6_PV(0, 0)
#undef SE3_X6_PV
      }
    }
    if (PROF && step < 5) { SE3_STAMP(5 + 5 * step) }
    publish(buf ^ 1, rk_pub, rv_pub);
    __syncthreads();
    if (PROF && step < 5) { SE3_STAMP(6 + 5 * step) }
  };
  // The step count is padded to a multiple of 3 (the padding steps see only masked keys: exp(-inf) = 0, the running maximum is finite after
  // the first real tile): with `if (step + 1 < steps)` around the second and third step the paths into the loop's back edge carry different
  // numbers of outstanding loads and the compiler waits vmcnt(0) there -- the full latency of the tile just requested, every third step.
  for (int step = 0; step < steps; step += 3) {                // set of tile t = t % 3
    one_step(step, rk[0], rv[0], rb[0], rs[0], rk[1], rv[1]);
    one_step(step + 1, rk[1], rv[1], rb[1], rs[1], rk[2], rv[2]);
    one_step(step + 2, rk[2], rv[2], rb[2], rs[2], rk[0], rv[0]);
  }
  SE3_STAMP(30)
  if (!active) return;
  const int nrow = n0 + c32;
  if (nrow < cl.N) {
    const float inv_l = 1.0f / st.l;
    float* op = p.out + a * p.o_sa + ((size_t)cl.q_start + nrow) * C + h * D;
    const float* vi = X.vinv + (size_t)ci * C + h * D;        // the values' channel scales leave here
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const float4 w = ld4(vi + 32 * dt + 8 * g + 4 * half);
        *reinterpret_cast<float4*>(op + 32 * dt + 8 * g + 4 * half) =
            make_float4(st.o[dt][4 * g] * inv_l * w.x, st.o[dt][4 * g + 1] * inv_l * w.y, st.o[dt][4 * g + 2] * inv_l * w.z,
                        st.o[dt][4 * g + 3] * inv_l * w.w);
      }
  }
}

// K / V^T pieces + the f16 kernel; false = this shape stays on attention_kernel (head dimension other than 64, unaligned rows, no workspace)
extern "C" size_t se3_attention_kv_pieces_bytes(int num_anchors, int64_t key_rows, int C, int v_row_stride) {
  if (num_anchors < 1 || key_rows < 1 || C < 8 || v_row_stride < 16) return 0;
  return (size_t)2 * num_anchors * ((size_t)key_rows * C + (size_t)C * v_row_stride) * sizeof(_Float16) + 256 +
         (x6_kinv_floats(num_anchors, C, key_rows) + x6_vinv_floats(C)) * sizeof(float);
}
static bool launch_attention_x6(AttnArgs& p, void* ws, size_t ws_bytes, hipStream_t st) {
  // (the plain cross-attention calls -- no logits, one or six small value sets -- stay on the f32 kernel: 16 us against 20 + 10 for the split)
  if (ws == nullptr || p.bias == nullptr || g_attn_variant == 11 || p.C / p.H != 64 || p.C % 8 || p.v_rs % 16 || p.q_rs % 4 || p.k_rs % 4 ||
      (reinterpret_cast<uintptr_t>(ws) & 15) || (reinterpret_cast<uintptr_t>(p.q) & 15) || (reinterpret_cast<uintptr_t>(p.k) & 15) ||
      (reinterpret_cast<uintptr_t>(p.v) & 15) || (p.A > 1 && (p.k_sa == 0 || p.v_sa == 0)))
    return false;
  int64_t R = 0;
  int nmax = 1;
  for (int c = 0; c < p.S.n; c++) {
    const StackCloud& cl = p.S.c[c];
    if (cl.k_start % 16) return false;
    R = cl.k_start + cl.M > R ? cl.k_start + cl.M : R;
    nmax = cl.N > nmax ? cl.N : nmax;
  }
  if (ws_bytes < se3_attention_kv_pieces_bytes(p.A, R, p.C, p.v_rs)) return false;
  const int64_t nk = (int64_t)p.A * R * (p.C / 8), nv = (int64_t)p.A * p.C * (p.v_rs / 8);      // uint4 per piece
  uint4* wk = static_cast<uint4*>(ws);
  uint4* wv = wk + 2 * nk;
  float* kinv = reinterpret_cast<float*>(wv + 2 * nv);
  float* vinv = kinv + x6_kinv_floats(p.A, p.C, R);
  int mmax = 1;
  for (int c = 0; c < p.S.n; c++) mmax = p.S.c[c].M > mmax ? p.S.c[c].M : mmax;
  X6SplitArgs sp{};
  sp.S = p.S; sp.A = p.A; sp.C = p.C; sp.H = p.H;
  sp.k = p.k; sp.vt = p.v; sp.k_rs = p.k_rs; sp.v_rs = p.v_rs; sp.k_sa = p.k_sa; sp.v_sa = p.v_sa; sp.k_rows = R;
  sp.outk = wk; sp.outv = wv; sp.kinv = kinv; sp.vinv = vinv;
  sp.q_groups = 0; sp.k_groups = (((mmax + 7) / 8) * p.H + 3) / 4;
  // (tag 3: the prologue of the attention launch that follows -- bench.py adds its time to that launch)
  launch_kernel(3, x6_split_kernel, dim3((unsigned)(p.A * sp.k_groups + (p.C + 3) / 4), (unsigned)p.S.n), dim3(256), st, sp);
  AttnPieces X;
  for (int pc = 0; pc < 2; pc++) {
    X.k[pc] = wk + pc * nk;
    X.v[pc] = wv + pc * nv;
  }
  X.kinv = kinv; X.vinv = vinv; X.nblk = (int)((R + 7) / 8);
  const dim3 grid((unsigned)((nmax + 127) / 128), (unsigned)p.H, (unsigned)(p.S.n * p.A));
  if (g_attn_variant == 12 && p.bias != nullptr) {
    p.prof = g_attn_prof;
    launch_kernel(2, attention_x6_kernel<true, true>, grid, dim3(256), st, p, X, (int64_t)R * p.C, (int64_t)p.C * p.v_rs);
  } else if (p.bias != nullptr) launch_kernel(2, attention_x6_kernel<true>, grid, dim3(256), st, p, X, (int64_t)R * p.C, (int64_t)p.C * p.v_rs);
  else launch_kernel(2, attention_x6_kernel<false>, grid, dim3(256), st, p, X, (int64_t)R * p.C, (int64_t)p.C * p.v_rs);
  return true;
}

static int launch_attention(AttnArgs& p, hipStream_t st) {
  int qt = 1;
  for (int c = 0; c < p.S.n; c++) qt = (p.S.c[c].N + 31) / 32 > qt ? (p.S.c[c].N + 31) / 32 : qt;
  p.QT = qt;
  p.G = p.S.n * p.A * p.H;
  dim3 grid((unsigned)(8 * ((p.G + 7) / 8) * qt));
  int rc = dispatch_head_dim(p.C / p.H, [&](auto d) {
    constexpr int D = decltype(d)::value;
    switch (g_attn_variant) {      // tuning hook
      case 1: launch_kernel(2, attention_kernel<D, 4, 2, 1>, grid, dim3(256), st, p); break;
      case 2: launch_kernel(2, attention_kernel<D, 4, 3, 0>, grid, dim3(256), st, p); break;
      case 3: launch_kernel(2, attention_kernel<D, 6, 2, 2>, grid, dim3(384), st, p); break;
      case 9: p.prof = g_attn_prof; launch_kernel(2, attention_kernel<D, 4, 3, 3>, grid, dim3(256), st, p); break;
      case 5:                                                                                     // f16 hi / lo split MFMAs (flash_tiles_f16)
        if constexpr (D % 16 == 0) launch_kernel(2, attention_kernel<D, 4, 3, 4>, grid, dim3(256), st, p);
        else launch_kernel(2, attention_kernel<D, 4, 3, 2>, grid, dim3(256), st, p);
        break;
      case 6: launch_kernel(2, attention_kernel<D, 2, 3, 2>, grid, dim3(128), st, p); break;
      case 7:
        if constexpr (D % 16 == 0) launch_kernel(2, attention_kernel<D, 2, 3, 4>, grid, dim3(128), st, p);
        else launch_kernel(2, attention_kernel<D, 2, 3, 2>, grid, dim3(128), st, p);
        break;
      case 8: launch_kernel(2, attention_kernel<D, 1, 3, 2>, grid, dim3(64), st, p); break;
      case 10:
        if constexpr (D % 16 == 0) launch_kernel(2, attention_kernel<D, 1, 3, 4>, grid, dim3(64), st, p);
        else launch_kernel(2, attention_kernel<D, 1, 3, 2>, grid, dim3(64), st, p);
        break;
      default: launch_kernel(2, attention_kernel<D, 4, 3, 2>, grid, dim3(256), st, p); break;
    }
  }, "attention");
  if (rc != SE3_OK) return rc;
  SE3_CHECK_LAUNCH("attention");
  return SE3_OK;
}

extern "C" int se3_attention_fwd(const float* q, const float* k, const float* v, const float* bias, int num_anchors, int N,
                                 int M, int C, int H, int q_row_stride, int k_row_stride, int v_row_stride,
                                 int64_t q_anchor_stride, int64_t k_anchor_stride, int64_t v_anchor_stride,
                                 int64_t out_anchor_stride, int bias_row_stride, float scale, float* out, void* stream) {
  SE3_REQUIRE(v_row_stride >= ((M + 31) / 32) * 32 && v_row_stride % 4 == 0, SE3_ERR_INVALID_ARG,
              "attention: transposed-value row stride must be a multiple of 4 covering ceil32(M)");
  SE3_REQUIRE(q_row_stride >= C && k_row_stride >= C && q_row_stride % 4 == 0 && k_row_stride % 4 == 0, SE3_ERR_INVALID_ARG,
              "attention: q/k row strides must be multiples of 4 and >= C");
  SE3_REQUIRE(q && k && v && out, SE3_ERR_INVALID_ARG, "attention: null pointer");
  SE3_REQUIRE(num_anchors >= 1 && N >= 1 && M >= 1 && H >= 1 && C % H == 0, SE3_ERR_INVALID_ARG, "attention: bad sizes");
  SE3_REQUIRE(bias == nullptr || (bias_row_stride >= ((M + 31) / 32) * 32 && bias_row_stride % 4 == 0), SE3_ERR_INVALID_ARG,
              "attention: bias rows must be padded to a multiple of 4 covering ceil32(M)");
  AttnArgs p{};
  p.q = q; p.k = k; p.v = v; p.bias = bias; p.out = out;
  p.S.n = 1;
  p.S.c[0] = StackCloud{nullptr, nullptr, 0, 0, N, M, bias_row_stride, 0, 0};
  p.C = C; p.H = H; p.A = num_anchors;
  p.q_rs = q_row_stride; p.k_rs = k_row_stride; p.v_rs = v_row_stride;
  p.q_sa = q_anchor_stride; p.k_sa = k_anchor_stride; p.v_sa = v_anchor_stride; p.o_sa = out_anchor_stride;
  p.scale = scale;
  return launch_attention(p, (hipStream_t)stream);
}

extern "C" int se3_attention_stack_fwd(const float* q, const float* k, const float* vt, const float* bias,
                                       const int64_t* q_starts, const int64_t* q_lengths, const int64_t* k_starts,
                                       const int64_t* k_lengths, const int64_t* bias_offsets, int num_clouds, int num_anchors,
                                       int C, int H, int q_row_stride, int k_row_stride, int v_row_stride,
                                       int64_t q_anchor_stride, int64_t k_anchor_stride, int64_t v_anchor_stride,
                                       int64_t out_anchor_stride, float scale, float* out, void* kv_pieces_workspace,
                                       size_t kv_pieces_bytes, void* stream) {
  SE3_REQUIRE(q && k && vt && out && q_starts && q_lengths && k_starts && k_lengths, SE3_ERR_INVALID_ARG,
              "attention_stack: null pointer");
  SE3_REQUIRE((bias == nullptr) == (bias_offsets == nullptr), SE3_ERR_INVALID_ARG,
              "attention_stack: bias and bias_offsets go together");
  SE3_REQUIRE(num_clouds >= 1 && num_clouds <= kMaxClouds, SE3_ERR_UNSUPPORTED, "attention_stack: %d clouds (1..%d)", num_clouds,
              kMaxClouds);
  SE3_REQUIRE(num_anchors >= 1 && H >= 1 && C % H == 0, SE3_ERR_INVALID_ARG, "attention_stack: bad sizes");
  SE3_REQUIRE(q_row_stride >= C && k_row_stride >= C && q_row_stride % 4 == 0 && k_row_stride % 4 == 0 && v_row_stride % 4 == 0,
              SE3_ERR_INVALID_ARG, "attention_stack: row strides must be multiples of 4 (q/k >= C)");
  AttnArgs p{};
  p.q = q; p.k = k; p.v = vt; p.bias = bias; p.out = out;
  p.S.n = num_clouds;
  for (int c = 0; c < num_clouds; c++) {
    const int N = (int)q_lengths[c], M = (int)k_lengths[c], Mp = ((M + 31) / 32) * 32;
    SE3_REQUIRE(N >= 1 && M >= 1 && q_starts[c] >= 0 && k_starts[c] >= 0 && k_starts[c] % 4 == 0 && k_starts[c] + Mp <= v_row_stride,
                SE3_ERR_INVALID_ARG, "attention_stack: cloud %d: key columns must start at a multiple of 4 and ceil32(M) fit the value rows", c);
    SE3_REQUIRE(bias == nullptr || (bias_offsets[c] >= 0 && bias_offsets[c] % 4 == 0), SE3_ERR_INVALID_ARG,
                "attention_stack: cloud %d logits offset", c);
    p.S.c[c] = StackCloud{nullptr, nullptr, (int)q_starts[c], (int)k_starts[c], N, M, Mp, 0, bias ? (long long)bias_offsets[c] : 0};
  }
  p.C = C; p.H = H; p.A = num_anchors;
  p.q_rs = q_row_stride; p.k_rs = k_row_stride; p.v_rs = v_row_stride;
  p.q_sa = q_anchor_stride; p.k_sa = k_anchor_stride; p.v_sa = v_anchor_stride; p.o_sa = out_anchor_stride;
  p.scale = scale;
  if (launch_attention_x6(p, kv_pieces_workspace, kv_pieces_bytes, (hipStream_t)stream)) {
    SE3_CHECK_LAUNCH("attention (f16 pieces)");
    return SE3_OK;
  }
  return launch_attention(p, (hipStream_t)stream);
}

static int rpe_self_attention_stack(const float* q, const float* k, const float* vt, const float* qp, const float* qe,
                                    int row_stride, int64_t anchor_stride, int v_row_stride, int64_t v_anchor_stride,
                                    const void* const* emb_ptrs, const float* const* eq_ptrs, const int64_t* starts,
                                    const int64_t* lengths, int num_clouds, int num_anchors, int C, int H,
                                    float* logits_workspace, int64_t out_anchor_stride, float* out, void* kv_pieces_workspace,
                                    size_t kv_pieces_bytes, void* stream, bool emb_bf16) {
  SE3_REQUIRE(starts && lengths && logits_workspace, SE3_ERR_INVALID_ARG, "rpe_self_attention_stack: null pointer");
  SE3_REQUIRE(num_clouds >= 1 && num_clouds <= kMaxClouds, SE3_ERR_UNSUPPORTED, "rpe_self_attention_stack: %d clouds (1..%d)",
              num_clouds, kMaxClouds);
  int64_t offsets[kMaxClouds];
  int64_t total = 0;
  for (int c = 0; c < num_clouds; c++) {
    offsets[c] = total;
    total += (int64_t)num_anchors * H * lengths[c] * (((lengths[c] + 31) / 32) * 32);
  }
  // the two timed launches of one call stay adjacent in the record even when several host threads launch
  std::unique_lock<std::recursive_mutex> lock(g_timed_mutex, std::defer_lock);
  if (g_time_kernels) {
    lock.lock();
    // algorithmic bytes of the call: q, k, v in + out, the embedding, the equivariant embedding
    double bytes = 0.0;
    for (int c = 0; c < num_clouds; c++) {
      const double n = (double)lengths[c];
      bytes += 4.0 * (4.0 * num_anchors * n * C + (qe ? num_anchors * n * n * 4.0 : 0.0)) + (emb_bf16 ? 2.0 : 4.0) * n * n * C;
    }
    g_next_aux = qe ? -bytes : bytes;
  }
  int rc = rpe_bias_stack(qp, qe, row_stride, anchor_stride, emb_ptrs, eq_ptrs, starts, lengths, lengths, offsets, num_clouds, C,
                          num_anchors * H, H, logits_workspace, stream, emb_bf16);
  if (rc != SE3_OK) return rc;
  return se3_attention_stack_fwd(q, k, vt, logits_workspace, starts, lengths, starts, lengths, offsets, num_clouds, num_anchors,
                                 C, H, row_stride, row_stride, v_row_stride, anchor_stride, anchor_stride, v_anchor_stride,
                                 out_anchor_stride, 1.0f / sqrtf((float)(C / H)), out, kv_pieces_workspace, kv_pieces_bytes, stream);
}

extern "C" int se3_rpe_self_attention_stack_fwd(const float* q, const float* k, const float* vt, const float* qp,
                                                const float* qe, int row_stride, int64_t anchor_stride, int v_row_stride,
                                                int64_t v_anchor_stride, const float* const* emb_ptrs,
                                                const float* const* eq_ptrs, const int64_t* starts, const int64_t* lengths,
                                                int num_clouds, int num_anchors, int C, int H, float* logits_workspace,
                                                int64_t out_anchor_stride, float* out, void* kv_pieces_workspace, size_t kv_pieces_bytes,
                                                void* stream) {
  return rpe_self_attention_stack(q, k, vt, qp, qe, row_stride, anchor_stride, v_row_stride, v_anchor_stride,
                                  reinterpret_cast<const void* const*>(emb_ptrs), eq_ptrs, starts, lengths, num_clouds,
                                  num_anchors, C, H, logits_workspace, out_anchor_stride, out, kv_pieces_workspace, kv_pieces_bytes, stream,
                                  false);
}

extern "C" int se3_rpe_self_attention_stack_bf16_fwd(const float* q, const float* k, const float* vt, const float* qp,
                                                     const float* qe, int row_stride, int64_t anchor_stride, int v_row_stride,
                                                     int64_t v_anchor_stride, const uint16_t* const* emb_ptrs,
                                                     const float* const* eq_ptrs, const int64_t* starts,
                                                     const int64_t* lengths, int num_clouds, int num_anchors, int C, int H,
                                                     float* logits_workspace, int64_t out_anchor_stride, float* out,
                                                     void* kv_pieces_workspace, size_t kv_pieces_bytes, void* stream) {
  SE3_REQUIRE(C % 32 == 0, SE3_ERR_UNSUPPORTED, "rpe_self_attention_stack_bf16: channels %d not a multiple of 32", C);
  return rpe_self_attention_stack(q, k, vt, qp, qe, row_stride, anchor_stride, v_row_stride, v_anchor_stride,
                                  reinterpret_cast<const void* const*>(emb_ptrs), eq_ptrs, starts, lengths, num_clouds,
                                  num_anchors, C, H, logits_workspace, out_anchor_stride, out, kv_pieces_workspace, kv_pieces_bytes, stream,
                                  true);
}

extern "C" int se3_cross_eq_stats(const float* q, const float* k, int A, int N, int M, int C, int H, float scale,
                                  float* partial, int* num_partials_per_pair, void* stream) {
  SE3_REQUIRE(q && k && partial && num_partials_per_pair, SE3_ERR_INVALID_ARG, "cross_eq_stats: null pointer");
  SE3_REQUIRE(A >= 1 && N >= 1 && M >= 1 && H >= 1 && C % H == 0, SE3_ERR_INVALID_ARG, "cross_eq_stats: bad sizes");
  dim3 grid((unsigned)((N + 31) / 32), (unsigned)(A * A));
  *num_partials_per_pair = (int)grid.x;
  hipStream_t st = (hipStream_t)stream;
  int rc = dispatch_head_dim(C / H, [&](auto d) {
    cross_eq_stats_kernel<decltype(d)::value><<<grid, 256, 0, st>>>(q, k, A, N, M, C, H, scale, partial);
  }, "cross_eq_stats");
  if (rc != SE3_OK) return rc;
  SE3_CHECK_LAUNCH("cross_eq_stats");
  return SE3_OK;
}

extern "C" int se3_cross_eq_mix(const float* partial, int num_partials_per_pair, int A, int N, int M, int mode,
                                const int64_t* trace_idx, int num_rotations, float* mix, float* weights, void* stream) {
  SE3_REQUIRE(partial && mix && weights, SE3_ERR_INVALID_ARG, "cross_eq_mix: null pointer");
  SE3_REQUIRE(A >= 1 && A * A <= 64 && (mode == 0 || (mode == 1 && trace_idx && num_rotations >= 1 && num_rotations <= 64)),
              SE3_ERR_UNSUPPORTED, "cross_eq_mix: A %d mode %d rotations %d", A, mode, num_rotations);
  cross_eq_mix_kernel<<<1, 64, 0, (hipStream_t)stream>>>(partial, num_partials_per_pair, 1.0f / ((float)N * (float)M), A,
                                                       num_rotations, trace_idx, mode, mix, weights, nullptr, Stack{}, 0);
  SE3_CHECK_LAUNCH("cross_eq_mix");
  return SE3_OK;
}

// Gram matrices of the packed rows of every (anchor, pair): G[a, p] = X_p^T X_p (C x C) with X_p = rows [starts[p], starts[p] + lengths[p]) of
// x[a] -- the operands of the anchor-pair statistics of the equivariant cross attention (se3_gram_frobenius).  v_mfma_f32_16x16x4_f32 with
// A = X^T read straight from the rows (A[i][k] = X[k][i]).  One workgroup per (a, p, 64 output rows), a wave per 64 columns.  Replaces
// index_select + mask + batched GEMM (three launches and a (A, P, W, C) copy per tensor).
namespace {
struct GramPairs {
  int n;
  int start[kMaxClouds], length[kMaxClouds];
};
template <int C>
__global__ __launch_bounds__(256) void gram_stack_kernel(const float* __restrict__ x, int64_t anchor_stride, GramPairs P, float* __restrict__ out) {
  // A wave owns a 64 x 64 block of G: lane (c, kq) loads ONE float4 of row k for the output rows (channels i0 + 4 c .. + 3) and one for the
  // columns (j0 + 4 c .. + 3), and the 16 MFMAs (ea, eb) of a K-step take component ea of the first and eb of the second: MFMA row m stands
  // for channel i0 + 4 m + ea, column n for j0 + 4 n + eb.  2 loads per 16 MFMAs (one dword per lane and MFMA operand took 5 loads per 4
  // MFMAs and was bound by L1 requests: 43-63 us per launch).
  constexpr int KS = 4;                                       // K-steps (4 rows each) per iteration
  const int ap = blockIdx.x, a = ap / P.n, p = ap - a * P.n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.y * 64, j0 = wave * 64;
  const int start = P.start[p], len = P.length[p];
  const float* xr = x + a * anchor_stride + (int64_t)start * C;
  f32x4 acc[4][4];
#pragma unroll
  for (int ea = 0; ea < 4; ea++)
#pragma unroll
    for (int eb = 0; eb < 4; eb++) acc[ea][eb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int k0, f32x4 (&av)[KS], f32x4 (&bv)[KS]) {
#pragma unroll
    for (int u = 0; u < KS; u++) {
      const int row = k0 + 4 * u + kq;
      const bool valid = row < len;
      const float* r = xr + (int64_t)(valid ? row : 0) * C;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      av[u] = valid ? *reinterpret_cast<const f32x4*>(r + i0 + 4 * c) : z;
      bv[u] = valid ? *reinterpret_cast<const f32x4*>(r + j0 + 4 * c) : z;
    }
  };
  f32x4 av[KS], bv[KS], an[KS], bn[KS];
  fetch(0, av, bv);
  for (int k0 = 0; k0 < len; k0 += 4 * KS) {
    fetch(k0 + 4 * KS, an, bn);                               // the next iteration's rows (past the end: zeros) behind this one's 64 MFMAs
#pragma unroll
    for (int u = 0; u < KS; u++)
#pragma unroll
      for (int ea = 0; ea < 4; ea++)
#pragma unroll
        for (int eb = 0; eb < 4; eb++) acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][ea], bv[u][eb], acc[ea][eb], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < KS; u++) {
      av[u] = an[u];
      bv[u] = bn[u];
    }
  }
  // acc[ea][eb][r] = G[i0 + 4 (4 kq + r) + ea][j0 + 4 c + eb]: the four eb of a lane are 16 contiguous bytes
  float* o = out + ((int64_t)ap * C + i0) * C + j0 + 4 * c;
#pragma unroll
  for (int ea = 0; ea < 4; ea++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      *reinterpret_cast<f32x4*>(o + (int64_t)(4 * (4 * kq + r) + ea) * C) = f32x4{acc[ea][0][r], acc[ea][1][r], acc[ea][2][r], acc[ea][3][r]};
}
// The same product in 32 x 32 blocks per wave (a float2 per lane and operand feeds 4 MFMAs): 4x as many waves with a quarter of the
// dependent MFMA chain each.  One pair per forward has 24 workgroups of the kernel above (232 compute units idle, 35 us: one wave's chain of
// 24 x 64 f32 MFMAs plus the exposed load latency); with a single wave per SIMD nothing else hides the L2 round trip, so the rows are
// requested three iterations (48 rows) ahead in a ring of four register sets.
template <int C>
__global__ __launch_bounds__(256) void gram_stack32_kernel(const float* __restrict__ x, int64_t anchor_stride, GramPairs P, float* __restrict__ out) {
  constexpr int KS = 4, NT = C / 32;
  const int ap = blockIdx.x, a = ap / P.n, p = ap - a * P.n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  // G is symmetric: only the NT (NT + 1) / 2 blocks with ti <= tj are computed, the others are their transposes (stored below)
  int t = blockIdx.y * 4 + wave, ti = 0;
  if (t >= NT * (NT + 1) / 2) return;
  while (t >= NT - ti) {
    t -= NT - ti;
    ti++;
  }
  const int tj = ti + t, i0 = ti * 32, j0 = tj * 32;
  const int start = P.start[p], len = P.length[p];
  const float* xr = x + a * anchor_stride + (int64_t)start * C;
  f32x4 acc[2][2];
#pragma unroll
  for (int ea = 0; ea < 2; ea++)
#pragma unroll
    for (int eb = 0; eb < 2; eb++) acc[ea][eb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int it, f32x2 (&av)[KS], f32x2 (&bv)[KS]) {           // unconditional loads (row clamped), rows past the end -> zeros
#pragma unroll
    for (int u = 0; u < KS; u++) {
      const int row = it * 4 * KS + 4 * u + kq;
      const float keep = row < len ? 1.f : 0.f;
      const float* r = xr + (int64_t)(row < len ? row : 0) * C;
      const f32x2 va = *reinterpret_cast<const f32x2*>(r + i0 + 2 * c), vb = *reinterpret_cast<const f32x2*>(r + j0 + 2 * c);
      av[u] = f32x2{va[0] * keep, va[1] * keep};
      bv[u] = vb;
    }
  };
  f32x2 ra[4][KS], rb[4][KS];
  fetch(0, ra[0], rb[0]);
  fetch(1, ra[1], rb[1]);
  fetch(2, ra[2], rb[2]);
  const int iters = (len + 4 * KS - 1) / (4 * KS);
  for (int it = 0; it < iters; it += 4) {                                 // (the last group may run up to 3 iterations on zeros)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      fetch(it + j + 3, ra[(j + 3) & 3], rb[(j + 3) & 3]);
#pragma unroll
      for (int u = 0; u < KS; u++)
#pragma unroll
        for (int ea = 0; ea < 2; ea++)
#pragma unroll
          for (int eb = 0; eb < 2; eb++)
            acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[j][u][ea], rb[j][u][eb], acc[ea][eb], 0, 0, 0);
    }
  }
  // acc[ea][eb][r] = G[i0 + 2 (4 kq + r) + ea][j0 + 2 c + eb]: the two eb of a lane are 8 contiguous bytes
  float* o = out + ((int64_t)ap * C + i0) * C + j0 + 2 * c;
#pragma unroll
  for (int ea = 0; ea < 2; ea++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      *reinterpret_cast<f32x2*>(o + (int64_t)(2 * (4 * kq + r) + ea) * C) = f32x2{acc[ea][0][r], acc[ea][1][r]};
  if (ti == tj) return;
  float* ot = out + ((int64_t)ap * C + j0 + 2 * c) * C + i0;             // G[j0 + 2 c + eb][i0 + 2 (4 kq + r) + ea]
#pragma unroll
  for (int eb = 0; eb < 2; eb++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      *reinterpret_cast<f32x2*>(ot + (int64_t)eb * C + 2 * (4 * kq + r)) = f32x2{acc[0][eb][r], acc[1][eb][r]};
}
}  // namespace

extern "C" int se3_gram_stack(const float* x, int num_anchors, int C, int64_t anchor_stride, const int64_t* starts, const int64_t* lengths,
                              int num_pairs, float* out, void* stream) {
  SE3_REQUIRE(x && starts && lengths && out, SE3_ERR_INVALID_ARG, "gram_stack: null pointer");
  SE3_REQUIRE(num_anchors >= 1 && num_pairs >= 1 && num_pairs <= kMaxClouds && (C == 128 || C == 256), SE3_ERR_UNSUPPORTED,
              "gram_stack: %d anchors, %d pairs (<= %d), C = %d (128 or 256)", num_anchors, num_pairs, kMaxClouds, C);
  GramPairs P{};
  P.n = num_pairs;
  for (int p = 0; p < num_pairs; p++) {
    SE3_REQUIRE(starts[p] >= 0 && lengths[p] >= 1, SE3_ERR_INVALID_ARG, "gram_stack: pair %d rows", p);
    P.start[p] = (int)starts[p];
    P.length[p] = (int)lengths[p];
  }
  // 32 x 32 blocks per wave up to 8 pairs (A = 6, C = 256: 14.7 / 15.4 / 25.5 us at 1 / 4 / 8 pairs against 34.6 / 35.0 / 36.9 us for the
  // 64 x 64 kernel, tools/micro/gram_tiles.py); beyond that the 64 x 64 kernel's fewer loads per MFMA win
  const bool small = num_anchors * num_pairs * (C / 64) <= 192;
  if (small) {
    const int NT = C / 32;
    const dim3 grid((unsigned)(num_anchors * num_pairs), (unsigned)((NT * (NT + 1) / 2 + 3) / 4));
    if (C == 256) gram_stack32_kernel<256><<<grid, 256, 0, (hipStream_t)stream>>>(x, anchor_stride, P, out);
    else gram_stack32_kernel<128><<<grid, 256, 0, (hipStream_t)stream>>>(x, anchor_stride, P, out);
    SE3_CHECK_LAUNCH("gram_stack");
    return SE3_OK;
  }
  const dim3 grid((unsigned)(num_anchors * num_pairs), (unsigned)(C / 64));
  if (C == 256) gram_stack_kernel<256><<<grid, 256, 0, (hipStream_t)stream>>>(x, anchor_stride, P, out);
  else gram_stack_kernel<128><<<grid, 128, 0, (hipStream_t)stream>>>(x, anchor_stride, P, out);
  SE3_CHECK_LAUNCH("gram_stack");
  return SE3_OK;
}

// <Gq[a, p], Gk[e, p]>_F for every (pair p, query anchor a, key anchor e): the anchor-pair statistics of the equivariant cross attention from
// the per-pair Gram matrices (se3_cross_eq_stack_fwd, sums_given).  gq, gk (A, P, n) contiguous; out (P, A, A) = factor * the inner products.
// One workgroup per (p, a, e): 2 n floats from L2 (each Gram matrix is read by A workgroups).  The library's batched GEMM took 51 us for
// this (A x n) . (n x A) shape with n = 65 536.
namespace {
__global__ __launch_bounds__(256) void gram_frobenius_kernel(const float* __restrict__ gq, const float* __restrict__ gk, int A, int P, int64_t n,
                                                             float factor, float* __restrict__ out) {
  __shared__ float red[4];
  const int e = blockIdx.x % A, a = (blockIdx.x / A) % A, p = blockIdx.x / (A * A);
  const float4* x = reinterpret_cast<const float4*>(gq + ((size_t)a * P + p) * n);
  const float4* y = reinterpret_cast<const float4*>(gk + ((size_t)e * P + p) * n);
  float acc = 0.f, acc2 = 0.f;
  int64_t i = threadIdx.x;
  for (; i + 768 < (n >> 2); i += 1024) {                     // 8 loads in flight per thread
    const float4 u0 = x[i], v0 = y[i], u1 = x[i + 256], v1 = y[i + 256], u2 = x[i + 512], v2 = y[i + 512], u3 = x[i + 768], v3 = y[i + 768];
    acc += u0.x * v0.x + u0.y * v0.y + u0.z * v0.z + u0.w * v0.w + u2.x * v2.x + u2.y * v2.y + u2.z * v2.z + u2.w * v2.w;
    acc2 += u1.x * v1.x + u1.y * v1.y + u1.z * v1.z + u1.w * v1.w + u3.x * v3.x + u3.y * v3.y + u3.z * v3.z + u3.w * v3.w;
  }
  for (; i < (n >> 2); i += 256) {
    const float4 u = x[i], v = y[i];
    acc += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
  }
  acc = se3_wave_sum(acc + acc2);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * factor;
}
}  // namespace

extern "C" int se3_gram_frobenius(const float* gq, const float* gk, int A, int num_pairs, int64_t elements, float factor, float* out,
                                  void* stream) {
  SE3_REQUIRE(gq && gk && out, SE3_ERR_INVALID_ARG, "gram_frobenius: null pointer");
  SE3_REQUIRE(A >= 1 && num_pairs >= 1 && elements >= 4 && elements % 4 == 0 && ((uintptr_t)gq & 15) == 0 && ((uintptr_t)gk & 15) == 0,
              SE3_ERR_UNSUPPORTED, "gram_frobenius: A %d pairs %d elements %lld (a multiple of 4, 16-byte aligned)", A, num_pairs,
              (long long)elements);
  gram_frobenius_kernel<<<(unsigned)(num_pairs * A * A), 256, 0, (hipStream_t)stream>>>(gq, gk, A, num_pairs, elements, factor, out);
  SE3_CHECK_LAUNCH("gram_frobenius");
  return SE3_OK;
}

extern "C" int se3_cross_eq_stack_fwd(const float* q, const float* k, const float* vt, const int64_t* q_starts,
                                      const int64_t* q_lengths, const int64_t* k_starts, const int64_t* k_lengths, int num_pairs,
                                      int A, int C, int H, int64_t q_anchor_stride, int64_t k_anchor_stride, int v_row_stride,
                                      int64_t v_anchor_stride, int mode, const int64_t* trace_idx, int num_rotations,
                                      int sums_given, float* partial_workspace, float* mix, float* weights, float* out,
                                      void* stream) {
  SE3_REQUIRE(q && k && vt && q_starts && q_lengths && k_starts && k_lengths && partial_workspace && mix && weights && out,
              SE3_ERR_INVALID_ARG, "cross_eq_stack: null pointer");
  SE3_REQUIRE(num_pairs >= 1 && num_pairs <= kMaxClouds, SE3_ERR_UNSUPPORTED, "cross_eq_stack: %d pairs (1..%d)", num_pairs, kMaxClouds);
  SE3_REQUIRE(A >= 1 && A * A <= 64 && H >= 1 && C % H == 0 && v_row_stride % 4 == 0, SE3_ERR_INVALID_ARG, "cross_eq_stack: bad sizes");
  SE3_REQUIRE(mode == 0 || (mode == 1 && trace_idx && num_rotations >= 1 && num_rotations <= 64), SE3_ERR_UNSUPPORTED,
              "cross_eq_stack: mode %d rotations %d", mode, num_rotations);
  CrossEqArgs p{};
  p.q = q; p.k = k; p.vt = vt;
  p.S.n = num_pairs;
  int qt = 1;
  for (int c = 0; c < num_pairs; c++) {
    const int N = (int)q_lengths[c], M = (int)k_lengths[c], Mp = ((M + 31) / 32) * 32;
    SE3_REQUIRE(N >= 1 && M >= 1 && q_starts[c] >= 0 && k_starts[c] >= 0 && k_starts[c] % 4 == 0 && k_starts[c] + Mp <= v_row_stride,
                SE3_ERR_INVALID_ARG, "cross_eq_stack: pair %d descriptor", c);
    p.S.c[c] = StackCloud{nullptr, nullptr, (int)q_starts[c], (int)k_starts[c], N, M, Mp, 0, 0};
    qt = (N + 31) / 32 > qt ? (N + 31) / 32 : qt;
  }
  p.A = A; p.C = C; p.H = H; p.QT = qt;
  p.q_sa = q_anchor_stride; p.k_sa = k_anchor_stride; p.v_sa = v_anchor_stride; p.v_rs = v_row_stride;
  p.scale = 1.0f / sqrtf((float)(C / H));
  hipStream_t st = (hipStream_t)stream;
  int rc = dispatch_head_dim(C / H, [&](auto d) {
    constexpr int D = decltype(d)::value;
    // sums_given: partial_workspace already holds sum_{n,m} (mean_h S[a,e,h,n,m])^2 per (pair, a, e) (the caller computed it from
    // Gram matrices: mean_h S = (scale / H) q_a[n].k_e[m] over all C channels, so the sum is (scale/H)^2 <Q_a^T Q_a, K_e^T K_e>)
    if (!sums_given)
      cross_eq_stats_stack_kernel<D><<<dim3((unsigned)qt, (unsigned)(A * A), (unsigned)num_pairs), 256, 0, st>>>(p, partial_workspace);
    cross_eq_mix_kernel<<<(unsigned)num_pairs, 64, 0, st>>>(partial_workspace, sums_given ? 1 : qt, 0.f, A, num_rotations, trace_idx, mode, mix,
                                                         weights, nullptr, p.S, 1);
    if (A <= 6)
      cross_eq_apply_stack_kernel<D, 6><<<dim3((unsigned)qt, (unsigned)H, (unsigned)(A * num_pairs)), 384, 0, st>>>(p, mix, out);
    else
      cross_eq_apply_stack_kernel<D, 4><<<dim3((unsigned)qt, (unsigned)H, (unsigned)(A * num_pairs)), 256, 0, st>>>(p, mix, out);
  }, "cross_eq_stack");
  if (rc != SE3_OK) return rc;
  SE3_CHECK_LAUNCH("cross_eq_stack");
  return SE3_OK;
}

// bf16x6 form of se3_cross_eq_stack_fwd (head dimension 64, A <= 6, key starts and the value row stride multiples of 16): q (A, q_rows, C),
// k (A, k_rows, C) and vt (A, C, v_row_stride) are split into f16 hi / lo pieces in `workspace` (se3_cross_eq_x6_workspace_bytes), then the flash
// loop runs on the bf16 matrix cores at f32 accuracy.  Other shapes take the f32 kernels of se3_cross_eq_stack_fwd.
extern "C" size_t se3_cross_eq_x6_workspace_bytes(int A, int64_t q_rows, int64_t k_rows, int C, int v_row_stride) {
  return (size_t)4 * A * ((size_t)q_rows * C + (size_t)k_rows * C + (size_t)C * v_row_stride) + 256 +      // two f16 pieces per value
         (x6_qinv_floats(A, C, q_rows) + x6_kinv_floats(A, C, k_rows) + x6_vinv_floats(C)) * sizeof(float);  // + their scales
}

extern "C" int se3_cross_eq_stack_x6_fwd(const float* q, const float* k, const float* vt, const int64_t* q_starts,
                                         const int64_t* q_lengths, const int64_t* k_starts, const int64_t* k_lengths, int num_pairs,
                                         int A, int C, int H, int64_t q_rows, int64_t k_rows, int64_t q_anchor_stride,
                                         int64_t k_anchor_stride, int v_row_stride, int64_t v_anchor_stride, int mode,
                                         const int64_t* trace_idx, int num_rotations, int sums_given, float* partial_workspace,
                                         float* mix, float* weights, float* out, int out_groups, int64_t out_anchor_stride,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(out_groups >= 1 && A >= 1 && A % out_groups == 0, SE3_ERR_INVALID_ARG, "cross_eq_stack_x6: %d groups of %d key anchors", out_groups, A);
  bool ok = workspace != nullptr && H >= 1 && C % H == 0 && C / H == 64 && A >= 1 && A <= 6 && v_row_stride % 16 == 0 && sums_given &&
            num_pairs >= 1 && num_pairs <= kMaxClouds && k_starts != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
  for (int c = 0; ok && c < num_pairs; c++) ok = k_starts[c] % 16 == 0;
  if (!ok) {
    SE3_REQUIRE(out_groups == 1 && out_anchor_stride == q_anchor_stride, SE3_ERR_UNSUPPORTED,
                "cross_eq_stack_x6: key-anchor groups / an output stride of its own need the f16 form (head dimension 64, aligned key starts)");
    return se3_cross_eq_stack_fwd(q, k, vt, q_starts, q_lengths, k_starts, k_lengths, num_pairs, A, C, H, q_anchor_stride, k_anchor_stride,
                                  v_row_stride, v_anchor_stride, mode, trace_idx, num_rotations, sums_given, partial_workspace, mix,
                                  weights, out, stream);
  }
  SE3_REQUIRE(q && k && vt && q_starts && q_lengths && k_lengths && partial_workspace && mix && weights && out, SE3_ERR_INVALID_ARG,
              "cross_eq_stack_x6: null pointer");
  SE3_REQUIRE(workspace_bytes >= se3_cross_eq_x6_workspace_bytes(A, q_rows, k_rows, C, v_row_stride), SE3_ERR_WORKSPACE,
              "cross_eq_stack_x6: workspace too small");
  SE3_REQUIRE(mode == 0 || (mode == 1 && trace_idx && num_rotations >= 1 && num_rotations <= 64), SE3_ERR_UNSUPPORTED,
              "cross_eq_stack_x6: mode %d rotations %d", mode, num_rotations);
  CrossEqArgs p{};
  p.q = q; p.k = k; p.vt = vt;
  p.S.n = num_pairs;
  int qt = 1;
  for (int c = 0; c < num_pairs; c++) {
    const int N = (int)q_lengths[c], M = (int)k_lengths[c], Mp = ((M + 31) / 32) * 32;
    SE3_REQUIRE(N >= 1 && M >= 1 && q_starts[c] >= 0 && k_starts[c] >= 0 && q_starts[c] + N <= q_rows && k_starts[c] + M <= k_rows &&
                    k_starts[c] + Mp <= v_row_stride,
                SE3_ERR_INVALID_ARG, "cross_eq_stack_x6: pair %d descriptor", c);
    p.S.c[c] = StackCloud{nullptr, nullptr, (int)q_starts[c], (int)k_starts[c], N, M, Mp, 0, 0};
    qt = (N + 31) / 32 > qt ? (N + 31) / 32 : qt;
  }
  p.A = A; p.C = C; p.H = H; p.QT = qt;
  p.q_sa = q_anchor_stride; p.k_sa = k_anchor_stride; p.v_sa = v_anchor_stride; p.v_rs = v_row_stride;
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.G = out_groups; p.out_rs = out_groups * C; p.out_sa = out_anchor_stride;
  hipStream_t st = (hipStream_t)stream;
  const size_t nq = (size_t)A * q_rows * C / 8, nk = (size_t)A * k_rows * C / 8, nv = (size_t)A * C * v_row_stride / 8;     // uint4 per piece
  uint4* wq = static_cast<uint4*>(workspace);
  uint4* wk = wq + 2 * nq;
  uint4* wv = wk + 2 * nk;
  float* qinv = reinterpret_cast<float*>(wv + 2 * nv);
  float* kinv = qinv + x6_qinv_floats(A, C, q_rows);
  float* vinv = kinv + x6_kinv_floats(A, C, k_rows);
  int nmax = 1, mmax = 1;
  for (int c = 0; c < num_pairs; c++) {
    nmax = p.S.c[c].N > nmax ? p.S.c[c].N : nmax;
    mmax = p.S.c[c].M > mmax ? p.S.c[c].M : mmax;
  }
  X6SplitArgs sp{};
  sp.S = p.S; sp.A = A; sp.C = C; sp.H = H;
  sp.q = q; sp.k = k; sp.vt = vt; sp.q_rs = C; sp.k_rs = C; sp.v_rs = v_row_stride;
  sp.q_sa = q_anchor_stride; sp.k_sa = k_anchor_stride; sp.v_sa = v_anchor_stride; sp.q_rows = q_rows; sp.k_rows = k_rows;
  sp.outq = wq; sp.outk = wk; sp.outv = wv; sp.qinv = qinv; sp.kinv = kinv; sp.vinv = vinv;
  sp.q_groups = (((nmax + 7) / 8) * H + 3) / 4; sp.k_groups = (((mmax + 7) / 8) * H + 3) / 4;
  x6_split_kernel<<<dim3((unsigned)(A * (sp.q_groups + sp.k_groups) + (C + 3) / 4), (unsigned)num_pairs), 256, 0, st>>>(sp);
  X6Pieces X;
  for (int pc = 0; pc < 2; pc++) {
    X.q[pc] = wq + pc * nq;
    X.k[pc] = wk + pc * nk;
    X.v[pc] = wv + pc * nv;
  }
  X.qinv = qinv; X.kinv = kinv; X.vinv = vinv; X.q_rows = q_rows; X.nblk = (int)((k_rows + 7) / 8);
  cross_eq_mix_kernel<<<(unsigned)num_pairs, 64, 0, st>>>(partial_workspace, 1, 0.f, A, num_rotations, trace_idx, mode, mix, weights, nullptr,
                                                       p.S, 1);
  cross_eq_apply_stack_x6_kernel<<<dim3((unsigned)((qt + 3) / 4), (unsigned)H, (unsigned)(A * num_pairs * out_groups)), 256, 0, st>>>(
      p, X, (int64_t)q_rows * C, (int64_t)k_rows * C, (int64_t)C * v_row_stride, mix, out);
  SE3_CHECK_LAUNCH("cross_eq_stack_x6");
  return SE3_OK;
}

extern "C" int se3_cross_eq_apply(const float* q, const float* k, const float* vt, const float* mix, int A, int N, int M, int C,
                                  int H, int key_stride, float scale, float* out, void* stream) {
  SE3_REQUIRE(key_stride >= ((M + 31) / 32) * 32 && key_stride % 4 == 0, SE3_ERR_INVALID_ARG, "cross_eq_apply: key stride");
  SE3_REQUIRE(q && k && vt && mix && out, SE3_ERR_INVALID_ARG, "cross_eq_apply: null pointer");
  SE3_REQUIRE(A >= 1 && N >= 1 && M >= 1 && H >= 1 && C % H == 0, SE3_ERR_INVALID_ARG, "cross_eq_apply: bad sizes");
  dim3 grid((unsigned)((N + 31) / 32), (unsigned)H, (unsigned)A);
  hipStream_t st = (hipStream_t)stream;
  int rc = dispatch_head_dim(C / H, [&](auto d) {
    cross_eq_apply_kernel<decltype(d)::value><<<grid, 256, 0, st>>>(q, k, vt, mix, A, N, M, C, key_stride, scale, out);
  }, "cross_eq_apply");
  if (rc != SE3_OK) return rc;
  SE3_CHECK_LAUNCH("cross_eq_apply");
  return SE3_OK;
}

extern "C" unsigned long long se3_debug_attention_saturated(int reset) {
  unsigned long long n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_attn_saturated), sizeof(n)) != hipSuccess) return ~0ull;
  if (reset) {
    const unsigned long long zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_saturated), &zero, sizeof(zero));
  }
  return n;
}
