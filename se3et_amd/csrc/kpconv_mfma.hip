// B1: E2PN anchor-group kernel-point convolution (KPConvInterSO3) on the matrix cores at f32 accuracy -- gather AND contraction.
//
// Reference: geotransformer/modules/e2pn/blocks_epn.py:334-390 (feat_gather_by_perm), :454-546 (forward), weight tables :228-332:
//   out[p, r, d] = sum_{(s, t), c} G[p, r, (s, t), c] W[s, t, c, d],      G[p, r, (s, t), c] = H[p, orbit(s, r), anchor(t, r), c]
//   H[p, o, a, c] = sum_n hw[p, n, o] x[idx[p, n], a, c],                  hw[p, n, o] = sum_{k in orbit o} max(0, 1 - |s_n - q_p - kp_k| / sigma)
// with the 16 kernel-point orbits of csrc/kpconv_sums.h.  Three stages, all MFMA:
//   1. kpconv_neighbor_table_kernel: per query point its VALID neighbours (compacted) and their 16 orbit weights hw (64 B per neighbour).
//   2. gather: per point and 16 input channels H = hw^T x is a (16 orbits x n) . (n x 96 columns) product: v_mfma_f32_16x16x4_f32 (exact f32
//      FMA chains) with the orbit weights as A and the gathered rows as B operand -- vector loads only (the round's first forms, 15 FMAs per
//      gathered element on the vector ALUs with the weights in LDS or, wave-uniform, in SGPRs through scalar loads, were bound by LDS reads
//      and by the scalar-load round trip per 6 rows that the SGPR file allows).  The result is split ONCE into f16 hi + lo pieces.
//   3. contraction: v_mfma_f32_32x32x16_f16, three products hi hi + hi lo + lo hi accumulated in f32 (2^-22 per term, below the f32 GEMM's own
//      accumulation error; the weights are scaled by a power of two so that their lo pieces stay normal numbers).  The K loop holds nothing
//      but ds_read_b128 (A: the tile image in LDS, read in place through a 72-entry offset table), weight-fragment loads (B: lane order, L1 /
//      L2 resident, requested ahead) and MFMAs.  A workgroup owns a 16-point tile = 96 output rows = three 32-row tiles (anchor pairs); the
//      row order inside a 32-row tile is chosen so that every 16-lane group of a ds_read_b128 reads ONE run of 16 different points:
//      conflict-free with the odd row stride of the image.
// Fused form (default): stages 2 and 3 in ONE kernel, producer waves filling the image of chunk t + 1 in LDS while consumer waves multiply
// chunk t: H never exists in HBM.  Two-launch form (se3_kpconv_so3_gather_sums + se3_kpconv_so3_contract_f16): the images go through HBM.
#include "common.h"
#include "kpconv_sums.h"

namespace {

using namespace kpsum;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;      // (an array of HIP uint4 structs ends up in scratch)
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Diagnostic hooks, empty in the product: tools/micro/kpconv_stamps.hip defines them (wave time stamps, fixed weight fragments) before it
// includes this file and builds a library of its own.
#ifndef SE3_STAMP
#define SE3_STAMP(step_, slot_)
#endif
#ifndef SE3_DIAG_WEIGHT_STEP
#define SE3_DIAG_WEIGHT_STEP(gs_, ksp_) (gs_)
#endif

constexpr int kTile4 = kTileB / 16;          // uint4 per tile image (3109)
constexpr int kHeaderB = 256;                // weight-fragment buffer: [header: 1 / scale, max |W| bits][fragments]

// ---- weights: (36 Cin, Cout) f32 -> f16 hi / lo fragments [chunk][K16-step][column tile][piece][lane] x 16 B, scaled by a power of two ----
__global__ void kpconv_wmax_kernel(const float* __restrict__ W, int64_t n, unsigned* __restrict__ hdr) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(W[i]));
  m = se3_wave_max(m);
  if (se3_lane() == 0) atomicMax(hdr + 1, __float_as_uint(m));      // non-negative floats order like their bit patterns
}

__device__ __forceinline__ float weight_scale(unsigned max_bits) {
  // power of two that brings max |W| into [2^12, 2^13): hi pieces far from f16 overflow, lo pieces (2^-11 below) normal down to |W| = max * 2^-15
  const int e = (int)((max_bits >> 23) & 0xff);                       // biased exponent of the maximum
  if (e == 0 || e == 0xff) return 1.f;
  return __uint_as_float((unsigned)(127 + 12 - (e - 127)) << 23);
}

__global__ void kpconv_split_weights_f16_kernel(const float* __restrict__ W, int Cin, int Cout, unsigned* __restrict__ hdr,
                                                uint4* __restrict__ Wf) {
  const int NCT = Cout / 32;
  const int64_t frag = blockIdx.x;                    // (chunk, K16-step, column tile)
  const int ct = (int)(frag % NCT), st = (int)((frag / NCT) % kSteps), cc = (int)(frag / ((int64_t)NCT * kSteps));
  const int lane = threadIdx.x, n = ct * 32 + (lane & 31), u = 2 * st + (lane >> 5);
  const float scale = weight_scale(hdr[1]);
  if (frag == 0 && lane == 0) reinterpret_cast<float*>(hdr)[0] = 1.f / scale;
  f16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const float w = W[((int64_t)u * Cin + cc * kCC + j) * Cout + n] * scale;
    hi[j] = (_Float16)w;
    lo[j] = (_Float16)(w - (float)hi[j]);
  }
  uint4* dst = Wf + frag * 2 * 64 + lane;
  dst[0] = __builtin_bit_cast(uint4, hi);
  dst[64] = __builtin_bit_cast(uint4, lo);
}

// ---- contraction -----------------------------------------------------------------------------------------------------------------------
// row i of a 32-row MFMA tile -> (point of the tile, anchor of the tile's pair): see the header (ds_read_b128 lane groups)
__device__ __forceinline__ int row_point(int i) { return ((i >> 3) << 2) + (i & 3); }
__device__ __forceinline__ int row_rsel(int i) { return (0x96 >> (i >> 2)) & 1; }

template <int NCW, int KS>      // column tiles (32 columns) over the waves; K16-steps of a chunk dealt over KS wave groups
__global__ __launch_bounds__(64 * NCW * KS) void kpconv_mfma_kernel(const u32x4* __restrict__ H, const u32x4* __restrict__ Wf,
                                                                    const float* __restrict__ hdr, int64_t P, int64_t tiles, int Cin,
                                                                    int Cout, float* __restrict__ out) {
  constexpr int kThreads = 64 * NCW * KS;
  constexpr int kPF = (kTile4 + kThreads - 1) / kThreads;              // uint4 of the tile image per thread
  constexpr int kSPW = kSteps / KS;                                    // K16-steps per wave and chunk
  static_assert(kSteps % KS == 0, "K split must divide the 18 K16-steps of a chunk");
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned* tab = reinterpret_cast<unsigned*>(lds + kTileB);           // [K16-step][rsel][h]: three 8-bit run numbers (anchor pairs 0..2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cw = wave % NCW, ksp = wave / NCW;
  const int64_t tile = blockIdx.x, p0 = tile * kTP;
  const int NCT = Cout / 32, ct = blockIdx.y * NCW + cw;
  for (int e = tid; e < kSteps * 4; e += kThreads) {
    const int h = e & 1, rsel = (e >> 1) & 1, st = e >> 2;
    const int u = 2 * st + h, s = u / kA, t = u % kA;
    unsigned v = 0;
    for (int rt = 0; rt < 3; rt++) {
      const int r = 2 * rt + rsel;
      v |= (unsigned)run_of(kOrb.id[s][r], kOrb.anchor[t][r]) << (8 * rt);
    }
    tab[e] = v;
  }
  const int i32 = lane & 31, h = lane >> 5;
  const int a_base = row_point(i32) * kRowB;                           // + piece * kPieceB + run * 16
  const int tab_lane = row_rsel(i32) * 2 + h;
  f32x16 acc[3];
#pragma unroll
  for (int rt = 0; rt < 3; rt++)
#pragma unroll
    for (int v = 0; v < 16; v++) acc[rt][v] = 0.f;
  const int chunks = Cin / kCC;
  // the tile image of chunk cc + 1 travels global -> registers while chunk cc is multiplied, registers -> LDS at the chunk boundary
  u32x4 pf[kPF];
  {
    const u32x4* src = H + tile * kTile4;
#pragma unroll
    for (int i = 0; i < kPF; i++) pf[i] = src[min(tid + i * kThreads, kTile4 - 1)];
  }
  // weight fragments of this wave's next K16-step (hi, lo), requested one step ahead
  const u32x4* wbase = Wf + (int64_t)ct * 2 * 64 + lane;
  const int64_t wstep = (int64_t)NCT * 2 * 64;                          // uint4 per K16-step over the whole layer
  u32x4 bn0 = wbase[(int64_t)ksp * wstep], bn1 = wbase[(int64_t)ksp * wstep + 64];
  const int64_t last_step = (int64_t)chunks * kSteps - KS + ksp;
  for (int cc = 0; cc < chunks; cc++) {
    __syncthreads();                                                    // the previous chunk's image is no longer read
#pragma unroll
    for (int i = 0; i < kPF; i++)
      if (tid + i * kThreads < kTile4) reinterpret_cast<u32x4*>(lds)[tid + i * kThreads] = pf[i];
    {
      const int cn = cc + 1 < chunks ? cc + 1 : cc;                      // unconditional (clamped) so that the compiler can count the requests
      const u32x4* src = H + ((int64_t)cn * tiles + tile) * kTile4;
#pragma unroll
      for (int i = 0; i < kPF; i++) pf[i] = src[min(tid + i * kThreads, kTile4 - 1)];
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < kSPW; q++) {
      const int st = ksp + q * KS;
      const unsigned runs = tab[st * 4 + tab_lane];
      f16x8 a[3][2];
#pragma unroll
      for (int rt = 0; rt < 3; rt++) {
        const int off = a_base + (int)((runs >> (8 * rt)) & 0xff) * 16;
        a[rt][0] = *reinterpret_cast<const f16x8*>(lds + off);
        a[rt][1] = *reinterpret_cast<const f16x8*>(lds + off + kPieceB);
      }
      const f16x8 b0 = __builtin_bit_cast(f16x8, bn0), b1 = __builtin_bit_cast(f16x8, bn1);
      {
        int64_t g = (int64_t)cc * kSteps + st + KS;
        g = g < last_step ? g : last_step;
        bn0 = wbase[g * wstep];
        bn1 = wbase[g * wstep + 64];
      }
      // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
      for (int rt = 0; rt < 3; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][1], b0, acc[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 3; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][0], b1, acc[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 3; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][0], b0, acc[rt], 0, 0, 0);
    }
  }
  if (KS > 1) {                                                          // merge the K split through LDS (the image is no longer needed)
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
    if (ksp > 0) {
#pragma unroll
      for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int v = 0; v < 16; v++) red[(((ksp - 1) * NCW + cw) * 48 + rt * 16 + v) * 64 + lane] = acc[rt][v];
    }
    __syncthreads();
    if (ksp > 0) return;
#pragma unroll
    for (int k = 1; k < KS; k++)
#pragma unroll
      for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[rt][v] += red[(((k - 1) * NCW + cw) * 48 + rt * 16 + v) * 64 + lane];
  }
  // accumulator register v of lane (column i32, half h): row (v & 3) + 8 (v >> 2) + 4 h of the 32-row tile = point v, anchor 2 rt + rsel
  const float inv_scale = hdr[0];
#pragma unroll
  for (int rt = 0; rt < 3; rt++)
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int64_t p = p0 + v;
      const int r = 2 * rt + ((0x96 >> (2 * (v >> 2) + h)) & 1);
      if (p < P) out[(p * kA + r) * Cout + ct * 32 + i32] = acc[rt][v] * inv_scale;
    }
}


// ---- stage 1: neighbour table ------------------------------------------------------------------------------------------------------------
// Per query point the VALID neighbours compacted to the front (invalid ones carry weight 0 in the reference: blocks_epn.py:471,377 shadow
// point / zero feature row), padded with (index 0, weights 0) up to a multiple of 32 entries (NNp):
//   hwt  [P][16][NNp] f32   orbit weights hw[o][n] = sum_{k in orbit o} max(0, 1 - |s_n - q - kp_k| / sigma)   (blocks_epn.py:520-533), orbit-major:
//                           a lane's 8 neighbours of one orbit are one 32-byte run
//   nbr  [P][NNp] int32     support row of the j-th valid neighbour
//   cnt  [P] int32          valid neighbours
__global__ __launch_bounds__(64) void kpconv_neighbor_table_kernel(const float* __restrict__ q_pts, const float* __restrict__ s_pts,
                                                                   const int64_t* __restrict__ idx, int64_t Ns, int NN, int NNp,
                                                                   const float* __restrict__ kp, float inv_sigma, float* __restrict__ hwt,
                                                                   int* __restrict__ nbr, int* __restrict__ cnt) {
  __shared__ int order[64];
  const int64_t p = blockIdx.x;
  const int n = threadIdx.x;
  const int64_t j = n < NN ? idx[p * NN + n] : -1;
  const bool valid = j >= 0 && j < Ns;
  const unsigned long long m = __ballot(valid);
  const int nv = __popcll(m);
  if (valid) order[__popcll(m & ((1ull << n) - 1ull))] = n;
  __syncthreads();
  if (n == 0) cnt[p] = nv;
  for (int e = n; e < NNp; e += 64) {
    float hw[kOrbits];
#pragma unroll
    for (int o = 0; o < kOrbits; o++) hw[o] = 0.f;
    int row = 0;
    if (e < nv) {
      const int64_t js = idx[p * NN + order[e]];
      row = (int)js;
      const float qx = q_pts[3 * p], qy = q_pts[3 * p + 1], qz = q_pts[3 * p + 2];
      const float sx = s_pts[3 * js], sy = s_pts[3 * js + 1], sz = s_pts[3 * js + 2];
      float w[kK];
#pragma unroll
      for (int k = 0; k < kK; k++) {
        const float dx = sx - qx - kp[3 * k], dy = sy - qy - kp[3 * k + 1], dz = sz - qz - kp[3 * k + 2];
        w[k] = fmaxf(0.f, 1.f - sqrtf(dx * dx + dy * dy + dz * dz) * inv_sigma);
      }
#pragma unroll
      for (int o = 0; o < kOrbits; o++)
#pragma unroll
        for (int k = 0; k < kK; k++)
          if ((kOrb.mask[o] >> k) & 1) hw[o] += w[k];
    }
    nbr[p * NNp + e] = row;
#pragma unroll
    for (int o = 0; o < kOrbits; o++) hwt[(p * kOrbits + o) * NNp + e] = hw[o];
  }
}

// ---- stage 2: gather on the f16 matrix cores (one wave = one point x 16 channels) ---------------------------------------------------------
// v_mfma_f32_16x16x32_f16: A[row = orbit l & 15][k = 8 (l >> 4) + j] = hw[orbit][neighbour 8 g + j], B[k][col = l & 15] = x[that neighbour][a][c0 + col],
// D[orbit 4 (l >> 4) + reg][col]: six column tiles (anchors) per point, 32 neighbours per MFMA; both operands as f16 hi + lo pieces, three
// products in f32 (2^-22 per term).  (The round's first MFMA gather used v_mfma_f32_16x16x4_f32 -- exact f32 -- at 1/16 of this rate: 36 MFMAs of
// 32 cycles per point and channel pair, as much matrix-pipe time as the contraction itself on the 64-wide layers; now 18 of 16 cycles plus
// ~300 vector instructions for the splits on otherwise idle ALUs.)
// The producers are bound by the number of vector-memory instructions, not by bytes: 8 waves x 58 requests per step take ~9.4 K cycles (one
// dword request per ~20 cycles and CU, the price of a 16-byte one), 2.8 K for splits + MFMAs, 2.4 K for the result split + stores (stamps of
// tools/micro/kpconv_stamps.py, layer 5); masking the lanes of k-groups without valid neighbours changed nothing.
// Neighbour slots per request round: 32 (one MFMA K-step: lane k-group g holds slots 8 g .. 8 g + 7) or, for tables wider than 32 (EXT), 40:
// slots 32 + 2 g, 32 + 2 g + 1 ride in a second K-step whose other six values per lane are zero.  At the bench shape the stage-2 / stage-3
// tables are 36 / 38 wide and 55-60 % of their points have more than 32 valid neighbours (tools/micro/neighbor_counts.py): a second,
// un-prefetched 32-slot round cost those points 6.8 K cycles per step (stamps), the two tail slots cost 12 more requests, and only for
// points that need them.
constexpr int kGN = 8, kGX = 2;

template <bool EXT>
struct GatherOps {                            // operands of one request round of one wave
  float aw[kGN + (EXT ? kGX : 0)];            // orbit weights (A): orbit l & 15, slots 8 g .. 8 g + 7 (, 32 + 2 g, 32 + 2 g + 1)
  float xb[kA][kGN + (EXT ? kGX : 0)];        // gathered feature values (B): column l & 15 of anchor a, the same slots
};

template <bool EXT>
__device__ __forceinline__ void gather_request_rows(const int* __restrict__ nbrow, int rd, int g, int (&nbv)[kGN + (EXT ? kGX : 0)]) {
  constexpr int S = EXT ? 40 : 32;
  const int4 lo = *reinterpret_cast<const int4*>(nbrow + S * rd + 8 * g), hi = *reinterpret_cast<const int4*>(nbrow + S * rd + 8 * g + 4);
  nbv[0] = lo.x; nbv[1] = lo.y; nbv[2] = lo.z; nbv[3] = lo.w; nbv[4] = hi.x; nbv[5] = hi.y; nbv[6] = hi.z; nbv[7] = hi.w;
  if constexpr (EXT) {
    const int2 t = *reinterpret_cast<const int2*>(nbrow + S * rd + 32 + 2 * g);
    nbv[8] = t.x;
    nbv[9] = t.y;
  }
}
// nv: valid neighbours of the point (uniform): the tail slots are requested only when the round has more than 32.
// BLK: x in the blocked layout [point][Cin / 16][anchor pair][16 channels][2 anchors] (rowops.hip: gn_chain_apply_blocked_kernel writes it):
// the six anchors of a lane's channel are three 8-byte loads, and one wave-instruction reads four neighbours x 128 contiguous bytes --
// whole cache lines.  In the plain layout (point, anchor, channel) an instruction reads four 64-byte half lines whose other halves the next
// step fetches again: twice the lines through the compute unit's L1, the producers' limit (tools/micro/kpconv_stamps.py).
template <bool EXT, bool BLK>
__device__ __forceinline__ void gather_request_ops(const float* __restrict__ x, const float* __restrict__ hwrow, int NNp, int rd, int g, int c16,
                                                   const int (&nbv)[kGN + (EXT ? kGX : 0)], unsigned rowlen, unsigned col, int Cin, int nv,
                                                   GatherOps<EXT>& q) {
  constexpr int S = EXT ? 40 : 32;
  const float* wr = hwrow + c16 * NNp + S * rd + 8 * g;
  const float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 4);
  q.aw[0] = w0.x; q.aw[1] = w0.y; q.aw[2] = w0.z; q.aw[3] = w0.w; q.aw[4] = w1.x; q.aw[5] = w1.y; q.aw[6] = w1.z; q.aw[7] = w1.w;
  const unsigned blk = (col >> 4) * 96 + (col & 15) * 2;                 // (BLK) floats from the point's row to this lane's anchor pair 0
  auto fetch = [&](int j) {
    if constexpr (BLK) {
#pragma unroll
      for (int ap = 0; ap < kA / 2; ap++) {
        const float2 v = *reinterpret_cast<const float2*>(x + (unsigned)nbv[j] * rowlen + blk + ap * 32);
        q.xb[2 * ap][j] = v.x;
        q.xb[2 * ap + 1][j] = v.y;
      }
    } else {
#pragma unroll
      for (int a = 0; a < kA; a++) q.xb[a][j] = x[(unsigned)nbv[j] * rowlen + (unsigned)(a * Cin) + col];
    }
  };
#pragma unroll
  for (int j = 0; j < kGN; j++) fetch(j);
  if constexpr (EXT) {
    if (nv > S * rd + 32) {
      const float2 wt = *reinterpret_cast<const float2*>(hwrow + c16 * NNp + S * rd + 32 + 2 * g);
      q.aw[8] = wt.x;
      q.aw[9] = wt.y;
#pragma unroll
      for (int j = kGN; j < kGN + kGX; j++) fetch(j);
    }
  }
}
__device__ __forceinline__ void split8(const float (&v)[kGN], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < kGN; i++) {
    hi[i] = (_Float16)v[i];
    lo[i] = (_Float16)(v[i] - (float)hi[i]);
  }
}
__device__ __forceinline__ void split2(float v0, float v1, f16x8& hi, f16x8& lo) {      // two values, six zeros
#pragma unroll
  for (int i = 0; i < 8; i++) hi[i] = lo[i] = (_Float16)0.f;
  hi[0] = (_Float16)v0; lo[0] = (_Float16)(v0 - (float)hi[0]);
  hi[1] = (_Float16)v1; lo[1] = (_Float16)(v1 - (float)hi[1]);
}
// tail: the round has more than 32 valid neighbours (uniform)
template <bool EXT>
__device__ __forceinline__ void gather_multiply(const GatherOps<EXT>& q, bool tail, f32x4 (&acc)[kA], float xs = 1.f) {      // xs: kpsum::x_split_scale
  f16x8 ah, al;
  {
    const float w8[kGN] = {q.aw[0], q.aw[1], q.aw[2], q.aw[3], q.aw[4], q.aw[5], q.aw[6], q.aw[7]};
    split8(w8, ah, al);
  }
#pragma unroll
  for (int a = 0; a < kA; a++) {
    const float v8[kGN] = {q.xb[a][0] * xs, q.xb[a][1] * xs, q.xb[a][2] * xs, q.xb[a][3] * xs, q.xb[a][4] * xs, q.xb[a][5] * xs, q.xb[a][6] * xs, q.xb[a][7] * xs};
    f16x8 bh, bl;
    split8(v8, bh, bl);
    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[a], 0, 0, 0);
    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[a], 0, 0, 0);
    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[a], 0, 0, 0);
  }
  if constexpr (EXT) {
    if (tail) {
      split2(q.aw[8], q.aw[9], ah, al);
#pragma unroll
      for (int a = 0; a < kA; a++) {
        f16x8 bh, bl;
        split2(q.xb[a][8] * xs, q.xb[a][9] * xs, bh, bl);
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[a], 0, 0, 0);
      }
    }
  }
}
// f16 hi / lo split of the wave's 24 values per lane, channel pairs exchanged inside lane pairs: even lanes end up with the hi dword of
// channels (c, c + 1), odd lanes with the lo dword of (c - 1, c); word[a][reg] belongs to run (a, orbit 4 g + reg)
__device__ __forceinline__ void gather_split(const f32x4 (&acc)[kA], int odd, unsigned (&word)[kA][4]) {
#pragma unroll
  for (int a = 0; a < kA; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float v = acc[a][r];
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      const unsigned u = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
      const unsigned w2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]: the pair's other lane
      word[a][r] = odd ? ((w2 >> 16) | (u & 0xffff0000u)) : ((u & 0xffffu) | (w2 << 16));
    }
}

// two-launch form: the images go to HBM.  One workgroup = one point x up to four channel pairs (one wave each); the rows are staged in
// wave-private LDS and leave as 16-byte stores.
constexpr int kStageRow[4] = {0, 392, 836, 1228};      // dword offsets of the (half, piece) rows of a wave's staging block: 4 banks apart
constexpr int kStageDw = 1228 + 384;
template <bool EXT>
__global__ __launch_bounds__(256) void kpconv_orbit_gather_kernel(const float* __restrict__ x, const float* __restrict__ hwt,
                                                                   const int* __restrict__ nbr, const int* __restrict__ cnt, int NNp,
                                                                   int64_t tiles, int Cin, unsigned char* __restrict__ Hs) {
  __shared__ __align__(16) unsigned stage[4][kStageDw];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15, odd = c16 & 1, half = c16 >> 3;
  const int64_t p = blockIdx.x;
  const int T = blockIdx.y * 4 + wave, chunks = Cin / kCC;
  if (2 * T >= chunks) return;
  const bool has2 = 2 * T + 1 < chunks;
  const unsigned rowlen = (unsigned)(kA * Cin), col = (unsigned)(T * 16 + (has2 || half == 0 ? c16 : c16 - 8));
  const int nv = cnt[p];
  const int* nbrow = nbr + p * NNp;
  const float* hwrow = hwt + p * NNp * 16;
  f32x4 acc[kA];
#pragma unroll
  for (int a = 0; a < kA; a++) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int S = EXT ? 40 : 32;
  for (int rd = 0; S * rd < nv; rd++) {
    int nbv[kGN + (EXT ? kGX : 0)];
    GatherOps<EXT> q;
    gather_request_rows<EXT>(nbrow, rd, g, nbv);
    gather_request_ops<EXT, false>(x, hwrow, NNp, rd, g, c16, nbv, rowlen, col, Cin, nv, q);
    gather_multiply<EXT>(q, nv > S * rd + 32, acc);
  }
  unsigned word[kA][4];
  gather_split(acc, odd, word);
  unsigned* st = stage[wave];
  const int cpair = (c16 & 7) >> 1;
#pragma unroll
  for (int a = 0; a < kA; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) st[kStageRow[half * 2 + odd] + (a * 16 + 4 * g + r) * 4 + cpair] = word[a][r];
  // (wave-private staging block: the LDS operations of a wave complete in order, no barrier)
  for (int e = lane; e < 4 * 96; e += 64) {
    const int row = e / 96, q4 = e - row * 96, hf = row >> 1, piece = row & 1;
    if (hf == 1 && !has2) continue;
    unsigned char* dst = Hs + ((int64_t)(2 * T + hf) * tiles + (p >> 4)) * kTileB + (size_t)piece * kPieceB + (size_t)(p & 15) * kRowB;
    reinterpret_cast<u32x4*>(dst)[q4] = *reinterpret_cast<const u32x4*>(st + kStageRow[row] + q4 * 4);
  }
}

// ---- fused form: stages 2 and 3 in one kernel ------------------------------------------------------------------------------------------
static int g_kpconv_variant = 0;
// One 11- or 12-wave workgroup per compute unit and 16-point tile; three tile images in LDS (3 x 48.6 KB).
// Schedule: step u = 0 .. chunks + 1, one barrier between steps.  CONSUMER waves multiply chunk u - 2 (image (u - 2) % 3) in step u >= 2.
// A PRODUCER wave (8 of them) handles ONE point per step over a PAIR of chunks (16 channels): in step u = 2T its first point, in step 2T + 1 its
// second, for chunks (2T, 2T + 1).  Only one image besides the one being filled is free for writing while three exist, so rows of chunk 2T go
// to image (2T) % 3 at once, while rows of chunk 2T + 1 computed in step 2T wait in registers for one step (image (2T + 1) % 3 is still being
// read during step 2T).  All operands of step u + 1 (neighbour numbers, orbit weights, 36 gathered values per lane) are requested during step
// u: the producers' memory latency (two dependent round trips, 1-3 us each under load) is off the critical path.
template <int NCW, int KS, int CT, bool EXT, bool BLK>      // consumer waves: NCW column groups x KS K-split groups; CT column tiles (32 columns) per wave; EXT: neighbour tables wider than 32; BLK: x in the blocked layout
__global__ __launch_bounds__(64 * (NCW * KS + 8)) void kpconv_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ hwt, const int* __restrict__ nbr, const int* __restrict__ cnt, int NNp,
    const u32x4* __restrict__ Wf, const float* __restrict__ hdr, int64_t P, int Cin, int Cout, float* __restrict__ out,
    float* __restrict__ split_part, int* __restrict__ split_count, int variant, const float* __restrict__ x_amax) {
  constexpr int NC = NCW * KS;                                         // consumer waves
  const float xs = x_split_scale(x_amax);                              // power-of-two scale of x before its f16 split (1 inside the plain range)
  constexpr int NPW = 8;                                               // producer waves: two points of the tile each
  constexpr int kSPW = kSteps / KS;                                    // K16-steps per consumer wave and chunk
  static_assert(kSteps % KS == 0, "K split must divide the 18 K16-steps of a chunk");
  extern __shared__ __align__(16) unsigned char lds[];                 // [3 images][table]
  unsigned* tab = reinterpret_cast<unsigned*>(lds + 3 * kTileB);       // [K16-step][rsel][h]: three 8-bit run numbers (anchor pairs 0..2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // variant bit 0: workgroup i runs on XCD i mod 8 -- give every XCD one contiguous range of tiles (its L2 then sees neighbouring tiles)
  int64_t tile = blockIdx.x;
  if (variant & 1) {
    const int n = (int)gridDim.x, xcd = (int)blockIdx.x & 7, per = n >> 3, rem = n & 7;
    tile = (int64_t)xcd * per + (xcd < rem ? xcd : rem) + ((int)blockIdx.x >> 3);
  }
  const int64_t p0 = tile * kTP;
  // gridDim.z > 1 (few tiles: one pair per forward): the input-channel chunks are split over gridDim.z workgroups, whose partial outputs
  // the last one to arrive adds up in a fixed order (see the epilogue)
  const int chunks = Cin / kCC / (int)gridDim.z, chunk0 = blockIdx.z * chunks, pairs = (chunks + 1) / 2;
  for (int e = tid; e < kSteps * 4; e += 64 * (NC + NPW)) {
    const int h = e & 1, rsel = (e >> 1) & 1, st = e >> 2;
    const int u = 2 * st + h, s = u / kA, t = u % kA;
    unsigned v = 0;
    for (int rt = 0; rt < 3; rt++) {
      const int r = 2 * rt + rsel;
      v |= (unsigned)run_of(kOrb.id[s][r], kOrb.anchor[t][r]) << (8 * rt);
    }
    tab[e] = v;
  }
  const int steps_total = chunks + 2;
  if (wave >= NC) {
    // ---------------- producer ----------------
    const int pw = wave - NC;
    const int g = lane >> 4, c16 = lane & 15, odd = c16 & 1, half = c16 >> 3, cpair = (c16 & 7) >> 1;
    const unsigned rowlen = (unsigned)(kA * Cin);
    // byte offset of this lane's dword of (anchor 0, orbit 4 g) inside its image (+ point * kRowB + (a * 16 + reg) * 16)
    const int dst0 = odd * kPieceB + g * 64 + cpair * 4;
    unsigned held[kA][4];
#pragma unroll
    for (int a = 0; a < kA; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) held[a][r] = 0u;
    // operands of the step about to run, requested one step ahead
    constexpr int S = EXT ? 40 : 32;
    GatherOps<EXT> ops;
    int nv_cur = 0;
    const int64_t plast = P - 1;
    auto point_of = [&](int u) { return p0 + pw + NPW * (u & 1); };
    auto col_of = [&](int u) {                                          // first of this lane's columns in step u (odd chunk count: the upper half re-reads the lower one)
      const int T = u >> 1;
      return (unsigned)(chunk0 * kCC + T * 16 + ((2 * T + 1 < chunks || half == 0) ? c16 : c16 - 8));
    };
    {
      const int64_t p = point_of(0), pc = p < P ? p : plast;
      int nbv[kGN + (EXT ? kGX : 0)];
      gather_request_rows<EXT>(nbr + pc * NNp, 0, g, nbv);
      nv_cur = p < P ? cnt[pc] : 0;
      gather_request_ops<EXT, BLK>(x, hwt + pc * NNp * 16, NNp, 0, g, c16, nbv, rowlen, col_of(0), Cin, __builtin_amdgcn_readfirstlane(nv_cur), ops);
    }
    for (int u = 0; u < steps_total; u++) {
      SE3_STAMP(u, 0)
      if (u < 2 * pairs) {
        const int T = u >> 1, second = u & 1;
        const int i = pw + NPW * second;                                  // point of the tile, uniform over the wave
        const int64_t p = p0 + i, pc = p < P ? p : plast;
        const bool has2 = 2 * T + 1 < chunks;                             // (odd chunk count: the last pair is a single chunk)
        unsigned char* img_a = lds + ((2 * T) % 3) * kTileB;
        unsigned char* img_b = lds + ((2 * T + 1) % 3) * kTileB;
        if (second && half == 1 && has2) {                                // the first point's rows of chunk 2T + 1, held since the last step
          unsigned char* dst = img_b + pw * kRowB + dst0;
#pragma unroll
          for (int a = 0; a < kA; a++)
#pragma unroll
            for (int r = 0; r < 4; r++) *reinterpret_cast<unsigned*>(dst + (a * 16 + r) * 16) = held[a][r];
        }
        // neighbour numbers and count of the NEXT step's point: in flight while this step multiplies
        const bool more = u + 1 < 2 * pairs;
        const int64_t pn = point_of(u + 1), pnc = pn < P ? pn : plast;
        int nbn[kGN + (EXT ? kGX : 0)];
        gather_request_rows<EXT>(nbr + pnc * NNp, 0, g, nbn);
        const int nv_next = (more && pn < P) ? cnt[pnc] : 0;
        SE3_STAMP(u, 1)
        f32x4 acc[kA];
#pragma unroll
        for (int a = 0; a < kA; a++) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nv = __builtin_amdgcn_readfirstlane(nv_cur);
        SE3_STAMP(u, 5)
        if (nv > 0) gather_multiply<EXT>(ops, nv > 32, acc, xs);
        SE3_STAMP(u, 6)
        for (int rd = 1; S * rd < nv; rd++) {                             // more valid neighbours than a round holds: further rounds, requested on the spot
          int nbv[kGN + (EXT ? kGX : 0)];
          GatherOps<EXT> q;
          gather_request_rows<EXT>(nbr + pc * NNp, rd, g, nbv);
          gather_request_ops<EXT, BLK>(x, hwt + pc * NNp * 16, NNp, rd, g, c16, nbv, rowlen, col_of(u), Cin, nv, q);
          gather_multiply<EXT>(q, nv > S * rd + 32, acc, xs);
        }
        SE3_STAMP(u, 7)
        // the next step's operands leave now (their neighbour numbers have arrived behind the MFMAs)
        gather_request_ops<EXT, BLK>(x, hwt + pnc * NNp * 16, NNp, 0, g, c16, nbn, rowlen, col_of(more ? u + 1 : u), Cin,
                                __builtin_amdgcn_readfirstlane(nv_next), ops);
        nv_cur = nv_next;
        SE3_STAMP(u, 2)
        unsigned word[kA][4];
        gather_split(acc, odd, word);
        unsigned char* dst = (half ? img_b : img_a) + i * kRowB + dst0;
        if (half == 0 || (second && has2)) {
#pragma unroll
          for (int a = 0; a < kA; a++)
#pragma unroll
            for (int r = 0; r < 4; r++) *reinterpret_cast<unsigned*>(dst + (a * 16 + r) * 16) = word[a][r];
        }
#pragma unroll
        for (int a = 0; a < kA; a++)
#pragma unroll
          for (int r = 0; r < 4; r++) held[a][r] = word[a][r];
      }
      SE3_STAMP(u, 3)
      if (u + 1 < steps_total) __syncthreads();
      SE3_STAMP(u, 4)
    }
    if (KS > 1) {
      __syncthreads();
      __syncthreads();
    }
    return;
  }
  // ---------------- consumer ----------------
  const int cw = wave % NCW, ksp = wave / NCW;
  const int NCT = Cout / 32, ct0 = (blockIdx.y * NCW + cw) * CT;
  const int i32 = lane & 31, h = lane >> 5;
  const int a_base = row_point(i32) * kRowB;                           // + piece * kPieceB + run * 16
  const int tab_lane = row_rsel(i32) * 2 + h;
  f32x16 acc[CT][3];
#pragma unroll
  for (int n = 0; n < CT; n++)
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[n][rt][v] = 0.f;
  // weight fragments (hi, lo) of this wave's next BD K16-steps: a ring of BD register sets, each refilled behind the MFMAs that read it
  // (with the fragments always L1-resident a chunk of the 256-wide layer takes 13.4 K cycles instead of 19.7 K: the L2 round trip of the
  // weight stream is the consumers' main stall, tools/micro/kpconv_stamps.py with -DSE3_DIAG_FIXED_B)
  constexpr int BD = (CT == 2 && KS > 1) ? 2 : (CT == 1 ? 6 : 3);        // ring depth (the K-split two-tile form has no registers for a third set;
                                                                        // one column tile per wave leaves room for six: the producers' branch sets the kernel's register count)
  constexpr int U = BD == 2 ? 2 : 6;                                    // unroll = lcm(2 A buffers, BD)
  static_assert(kSPW % U == 0, "K16-steps per wave and chunk must be a multiple of the unroll");
  const int64_t wstep = (int64_t)NCT * 2 * 64;                          // uint4 per K16-step over the whole layer
  const u32x4* wbase = Wf + (int64_t)chunk0 * kSteps * wstep + (int64_t)ct0 * 2 * 64 + lane;
  const int64_t last_step = (int64_t)chunks * kSteps - KS + ksp;
  u32x4 bq[BD][CT][2];
#pragma unroll
  for (int j = 0; j < BD; j++) {
    int64_t gs = ksp + j * KS;
    gs = gs < last_step ? gs : last_step;
#pragma unroll
    for (int n = 0; n < CT; n++) {
      bq[j][n][0] = wbase[gs * wstep + n * 128];
      bq[j][n][1] = wbase[gs * wstep + n * 128 + 64];
    }
  }
  __syncthreads();                                                      // steps 0 and 1: nothing to multiply yet
  __syncthreads();
  for (int cc = 0; cc < chunks; cc++) {
    const unsigned char* img = lds + (cc % 3) * kTileB;
    SE3_STAMP(cc + 2, 0)
    // A fragments of the step after the one being multiplied are read while its MFMAs run (two register sets)
    f16x8 av[2][3][2];
    {
      const unsigned runs = tab[ksp * 4 + tab_lane];
#pragma unroll
      for (int rt = 0; rt < 3; rt++) {
        const int off = a_base + (int)((runs >> (8 * rt)) & 0xff) * 16;
        av[0][rt][0] = *reinterpret_cast<const f16x8*>(img + off);
        av[0][rt][1] = *reinterpret_cast<const f16x8*>(img + off + kPieceB);
      }
    }
#pragma unroll 1
    for (int q0 = 0; q0 < kSPW; q0 += U) {
#pragma unroll
      for (int j = 0; j < U; j++) {
        constexpr int kDummy = 0;
        const int ja = j & 1, jb = j % BD;
        const int st = ksp + (q0 + j) * KS;
        {
          const int sn = st + KS < kSteps ? st + KS : st;                // next step of this chunk (clamped: the last one re-reads itself)
          const unsigned runs = tab[sn * 4 + tab_lane];
#pragma unroll
          for (int rt = 0; rt < 3; rt++) {
            const int off = a_base + (int)((runs >> (8 * rt)) & 0xff) * 16;
            av[ja ^ 1][rt][0] = *reinterpret_cast<const f16x8*>(img + off);
            av[ja ^ 1][rt][1] = *reinterpret_cast<const f16x8*>(img + off + kPieceB);
          }
          // (round 5, late: the six reads stay HERE, a whole K-step ahead of their use -- left alone the scheduler sinks each ds_read_b128 to
          // the MFMA that consumes it and the MFMA then waits out the LDS latency: the consumers lost 18 % to it)
          if constexpr (CT == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b0 = __builtin_bit_cast(f16x8, bq[jb][n][0]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ja][rt][1], b0, acc[n][rt], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b1 = __builtin_bit_cast(f16x8, bq[jb][n][1]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ja][rt][0], b1, acc[n][rt], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b0 = __builtin_bit_cast(f16x8, bq[jb][n][0]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ja][rt][0], b0, acc[n][rt], 0, 0, 0);
        }
        {
          int64_t gs = (int64_t)cc * kSteps + st + BD * KS;              // unconditional (clamped) so that the compiler can count the requests
          gs = gs < last_step ? gs : last_step;
          gs = SE3_DIAG_WEIGHT_STEP(gs, ksp);
#pragma unroll
          for (int n = 0; n < CT; n++) {
            bq[jb][n][0] = wbase[gs * wstep + n * 128];
            bq[jb][n][1] = wbase[gs * wstep + n * 128 + 64];
          }
        }
        (void)kDummy;
      }
    }
    SE3_STAMP(cc + 2, 3)
    if (cc + 1 < chunks) __syncthreads();
    SE3_STAMP(cc + 2, 4)
  }
  if (KS > 1) {                                                          // merge the K split through LDS (the images are no longer needed)
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
    if (ksp > 0) {
#pragma unroll
      for (int n = 0; n < CT; n++)
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
          for (int v = 0; v < 16; v++) red[((((ksp - 1) * NCW + cw) * CT + n) * 48 + rt * 16 + v) * 64 + lane] = acc[n][rt][v];
    }
    __syncthreads();
    if (ksp > 0) return;
#pragma unroll 1
    for (int k = 1; k < KS; k++)
#pragma unroll
      for (int n = 0; n < CT; n++)
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
          for (int v = 0; v < 16; v++) acc[n][rt][v] += red[((((k - 1) * NCW + cw) * CT + n) * 48 + rt * 16 + v) * 64 + lane];
  }
  const float inv_scale = hdr[0] / xs;
  // Output addressing: row (point v, rotation r = 2 rt + (h ^ s[v >> 2])), s = 0, 1, 1, 0 -- the accumulator's row order; buffer stores with
  // the tile as the buffer (rows past the last point fall outside and are dropped), lane part in two offsets, the rest uniform.
  const int row_b = Cout * 4;
  const int rows_here = (int)(P - p0 < kTP ? P - p0 : kTP);
  int voff[CT][2];
#pragma unroll
  for (int n = 0; n < CT; n++) {
    voff[n][0] = ((ct0 + n) * 32 + i32) * 4 + h * row_b;
    voff[n][1] = ((ct0 + n) * 32 + i32) * 4 + (1 - h) * row_b;
  }
  constexpr int kAgent = 16;                                              // sc1: agent scope (the access goes through to the memory side)
#pragma unroll
  for (int n = 0; n < CT; n++)
#pragma unroll
    for (int rt = 0; rt < 3; rt++) acc[n][rt] *= inv_scale;
  if (gridDim.z > 1) {
    // Split input channels: this wave's slice of the tile goes to the partial buffer of its split; the wave that finds itself last of the
    // gridDim.z that own the slice (arrival counter, reset for the next launch) adds the partials in split order and writes the output:
    // no atomics on the output, the same sum whatever the arrival order.  (Agent-scope stores / loads; no fence: see group_norm.h.)
    const int64_t rows_all = (int64_t)gridDim.x * kTP * kA;
    {
      const __amdgpu_buffer_rsrc_t prs =
          __builtin_amdgcn_make_buffer_rsrc(split_part + (blockIdx.z * rows_all + p0 * kA) * Cout, 0, kTP * kA * row_b, 0x00020000);
#pragma unroll
      for (int n = 0; n < CT; n++)
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
          for (int v = 0; v < 16; v++) {
            const float val = acc[n][rt][v];          // (a scalar copy: __builtin_bit_cast of a vector ELEMENT reads element 0, hipcc 7.2)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), prs, voff[n][(0x6 >> (v >> 2)) & 1],
                                                  (v * kA + 2 * rt) * row_b, kAgent);
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int* counter = split_count + (int64_t)tile * (NCT / CT) + ct0 / CT;
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != (int)gridDim.z - 1) return;
    if (lane == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int n = 0; n < CT; n++)
#pragma unroll
      for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[n][rt][v] = 0.f;
#pragma unroll 1
    for (int z = 0; z < (int)gridDim.z; z++) {
      const __amdgpu_buffer_rsrc_t prs =
          __builtin_amdgcn_make_buffer_rsrc(split_part + (z * rows_all + p0 * kA) * Cout, 0, kTP * kA * row_b, 0x00020000);
#pragma unroll
      for (int n = 0; n < CT; n++)
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
          for (int v = 0; v < 16; v++)
            acc[n][rt][v] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, voff[n][(0x6 >> (v >> 2)) & 1],
                                                                                            (v * kA + 2 * rt) * row_b, kAgent));
    }
  }
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out + p0 * kA * Cout, 0, rows_here * kA * row_b, 0x00020000);
#pragma unroll
  for (int n = 0; n < CT; n++)
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const float val = acc[n][rt][v];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ors, voff[n][(0x6 >> (v >> 2)) & 1], (v * kA + 2 * rt) * row_b, 0);
      }
}

// KPConvInterSO3.forward (blocks_epn.py:454-546) after se3_kpconv_neighbor_table: gather + contraction in one kernel.
}  // namespace

extern "C" size_t se3_kpconv_sums_bytes(int64_t num_queries, int in_channels) {
  if (num_queries < 0 || in_channels <= 0 || in_channels % kCC) return 0;
  return (size_t)(in_channels / kCC) * (size_t)se3_cdiv(num_queries, kTP) * kTileB;
}

extern "C" size_t se3_kpconv_weight_pieces_bytes(int in_channels, int out_channels) {
  if (in_channels <= 0 || out_channels <= 0 || in_channels % kCC || out_channels % 32) return 0;
  return kHeaderB + (size_t)(in_channels / kCC) * kSteps * (out_channels / 32) * 2 * 64 * sizeof(uint4);
}

extern "C" int se3_kpconv_split_weights_f16(const float* weights, int in_channels, int out_channels, void* pieces, void* stream) {
  SE3_REQUIRE(weights && pieces, SE3_ERR_INVALID_ARG, "kpconv_split_weights_f16: null pointer");
  SE3_REQUIRE(in_channels > 0 && in_channels % kCC == 0 && out_channels > 0 && out_channels % 32 == 0, SE3_ERR_UNSUPPORTED,
              "kpconv_split_weights_f16: channels (%d, %d) must be multiples of (8, 32)", in_channels, out_channels);
  hipStream_t st = (hipStream_t)stream;
  unsigned* hdr = static_cast<unsigned*>(pieces);
  if (hipMemsetAsync(hdr, 0, kHeaderB, st) != hipSuccess) {
    se3_set_error("kpconv_split_weights_f16: memset failed");
    return SE3_ERR_LAUNCH;
  }
  const int64_t n = (int64_t)kS * kA * in_channels * out_channels;
  kpconv_wmax_kernel<<<(unsigned)(n / 4096 < 1 ? 1 : (n / 4096 > 1024 ? 1024 : n / 4096)), 256, 0, st>>>(weights, n, hdr);
  const int64_t frags = (int64_t)(in_channels / kCC) * kSteps * (out_channels / 32);
  kpconv_split_weights_f16_kernel<<<(unsigned)frags, 64, 0, st>>>(
      weights, in_channels, out_channels, hdr, reinterpret_cast<uint4*>(static_cast<unsigned char*>(pieces) + kHeaderB));
  SE3_CHECK_LAUNCH("kpconv_split_weights_f16");
  return SE3_OK;
}

extern "C" int se3_kpconv_so3_contract_f16(const void* sums, const void* weight_pieces, int64_t num_queries, int in_channels,
                                           int out_channels, float* out, void* stream) {
  SE3_REQUIRE(sums && weight_pieces && out, SE3_ERR_INVALID_ARG, "kpconv_so3_contract_f16: null pointer");
  SE3_REQUIRE(in_channels > 0 && in_channels % kCC == 0 && out_channels >= 32 && out_channels % 32 == 0, SE3_ERR_UNSUPPORTED,
              "kpconv_so3_contract_f16: channels (%d, %d) must be multiples of (8, 32)", in_channels, out_channels);
  if (num_queries == 0) return SE3_OK;
  const int NCT = out_channels / 32;
  const int64_t tiles = se3_cdiv(num_queries, kTP);
  const u32x4* H = static_cast<const u32x4*>(sums);
  const float* hdr = static_cast<const float*>(weight_pieces);
  const u32x4* Wf = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces) + kHeaderB);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)kTileB + kSteps * 4 * sizeof(unsigned);
  if (NCT % 4 == 0) {
    kpconv_mfma_kernel<4, 1><<<dim3((unsigned)tiles, (unsigned)(NCT / 4)), 256, lds, st>>>(H, Wf, hdr, num_queries, tiles, in_channels,
                                                                                             out_channels, out);
  } else if (NCT % 2 == 0) {
    kpconv_mfma_kernel<2, 2><<<dim3((unsigned)tiles, (unsigned)(NCT / 2)), 256, lds, st>>>(H, Wf, hdr, num_queries, tiles, in_channels,
                                                                                             out_channels, out);
  } else {
    kpconv_mfma_kernel<1, 3><<<dim3((unsigned)tiles, (unsigned)NCT), 192, lds, st>>>(H, Wf, hdr, num_queries, tiles, in_channels,
                                                                                       out_channels, out);
  }
  SE3_CHECK_LAUNCH("kpconv_so3_contract_f16");
  return SE3_OK;
}

extern "C" size_t se3_kpconv_neighbor_table_bytes(int64_t num_queries, int num_neighbors) {
  if (num_queries < 0 || num_neighbors < 1 || num_neighbors > 64) return 0;
  const size_t nnp = num_neighbors <= 32 ? 32 : (size_t)(num_neighbors + 39) / 40 * 40;
  return (size_t)num_queries * nnp * (16 * sizeof(float) + sizeof(int)) + (size_t)num_queries * sizeof(int) + 256;
}

namespace {
struct NeighborTable {
  const float* hwt;
  const int *nbr, *cnt;
  int NNp;
};
NeighborTable table_views(const void* table, int64_t P, int NN) {
  NeighborTable t;
  t.NNp = NN <= 32 ? 32 : (NN + 39) / 40 * 40;      // rounds of 32 slots, or of 40 for tables wider than 32 (csrc: GatherOps<EXT>)
  t.hwt = static_cast<const float*>(table);
  t.nbr = reinterpret_cast<const int*>(t.hwt + (size_t)P * t.NNp * 16);
  t.cnt = t.nbr + (size_t)P * t.NNp;
  return t;
}
}  // namespace

extern "C" int se3_kpconv_neighbor_table(const float* q_pts, const float* s_pts, const int64_t* idx, const float* kernel_points_dev,
                                         float sigma, int64_t num_queries, int64_t num_support, int num_neighbors, void* table,
                                         size_t table_bytes, void* stream) {
  SE3_REQUIRE(q_pts && s_pts && idx && kernel_points_dev && table, SE3_ERR_INVALID_ARG, "kpconv_neighbor_table: null pointer");
  SE3_REQUIRE(num_neighbors >= 1 && num_neighbors <= 64 && sigma > 0.f, SE3_ERR_UNSUPPORTED, "kpconv_neighbor_table: %d neighbours (max 64)",
              num_neighbors);
  SE3_REQUIRE(num_support < (1ll << 31), SE3_ERR_UNSUPPORTED, "kpconv_neighbor_table: more than 2^31 support points");
  SE3_REQUIRE(table_bytes >= se3_kpconv_neighbor_table_bytes(num_queries, num_neighbors), SE3_ERR_INVALID_ARG,
              "kpconv_neighbor_table: table buffer too small");
  if (num_queries == 0) return SE3_OK;
  const NeighborTable t = table_views(table, num_queries, num_neighbors);
  kpconv_neighbor_table_kernel<<<(unsigned)num_queries, 64, 0, (hipStream_t)stream>>>(
      q_pts, s_pts, idx, num_support, num_neighbors, t.NNp, kernel_points_dev, 1.0f / sigma, const_cast<float*>(t.hwt),
      const_cast<int*>(t.nbr), const_cast<int*>(t.cnt));
  SE3_CHECK_LAUNCH("kpconv_neighbor_table");
  return SE3_OK;
}

extern "C" int se3_kpconv_so3_gather_sums(const float* x, const void* table, int64_t num_queries, int64_t num_support, int num_neighbors,
                                          int in_channels, void* sums, void* stream) {
  SE3_REQUIRE(x && table && sums, SE3_ERR_INVALID_ARG, "kpconv_so3_gather_sums: null pointer");
  SE3_REQUIRE(num_neighbors >= 1 && num_neighbors <= 64, SE3_ERR_UNSUPPORTED, "kpconv_so3_gather_sums: %d neighbours (max 64)", num_neighbors);
  SE3_REQUIRE(in_channels >= 8 && in_channels % 8 == 0, SE3_ERR_UNSUPPORTED, "kpconv_so3_gather_sums: channels must be a multiple of 8");
  SE3_REQUIRE((int64_t)num_support * kA * in_channels < (1ll << 31), SE3_ERR_UNSUPPORTED, "kpconv_so3_gather_sums: support features exceed 2^31 elements");
  if (num_queries == 0) return SE3_OK;
  const NeighborTable t = table_views(table, num_queries, num_neighbors);
  const int pairs = (in_channels / kCC + 1) / 2, waves = pairs < 4 ? pairs : 4;
  const dim3 grid((unsigned)num_queries, (unsigned)((pairs + 3) / 4));
  if (t.NNp > 32)
    kpconv_orbit_gather_kernel<true><<<grid, 64 * waves, 0, (hipStream_t)stream>>>(x, t.hwt, t.nbr, t.cnt, t.NNp, se3_cdiv(num_queries, kTP),
                                                                                  in_channels, static_cast<unsigned char*>(sums));
  else
    kpconv_orbit_gather_kernel<false><<<grid, 64 * waves, 0, (hipStream_t)stream>>>(x, t.hwt, t.nbr, t.cnt, t.NNp, se3_cdiv(num_queries, kTP),
                                                                                   in_channels, static_cast<unsigned char*>(sums));
  SE3_CHECK_LAUNCH("kpconv_so3_gather_sums");
  return SE3_OK;
}

// KPConvInterSO3.forward (blocks_epn.py:454-546) after se3_kpconv_neighbor_table: gather + contraction in one kernel.
// Few tiles (one pair per forward, the coarse stages): the input-channel chunks of a tile are split over `splits` workgroups.
// One workgroup per compute unit is resident (12 waves, 149 KB of LDS), so a launch runs in ceil(workgroups / 256) rounds, and a workgroup
// lasts (chunks / z + 3) steps when the input channels are split over z workgroups (two steps of pipeline fill, one of reduction).  The
// z with the smallest rounds x steps wins when it saves 10 % or more: 44 tiles (stage 3 of one pair, 32 chunks) -> z = 4; 345 tiles (stage
// 3 of the 8-pair batch: 2 rounds, the second one-third full) -> z = 2 (3 rounds of 19 steps instead of 2 of 35).
static int fused_splits(int64_t tiles, int colblocks, int chunks) {
  const int64_t wg = tiles * colblocks;
  int best = 1;
  int64_t best_cost = ((wg + 255) / 256) * (chunks + 3);
  for (int z = 2; z <= 8; z *= 2) {
    if (chunks % (2 * z) != 0 || chunks / z < 4) break;          // an even number of chunks per workgroup, at least 4
    const int64_t cost = ((wg * z + 255) / 256) * (chunks / z + 3);
    if (10 * cost <= 9 * best_cost) {
      best = z;
      best_cost = cost;
    }
  }
  return best;
}
static int64_t fused_tiles(int64_t num_queries) { return se3_cdiv(num_queries, kTP); }
static int fused_colblocks(int out_channels) {
  const int NCT = out_channels / 32;
  return NCT % 8 == 0 ? NCT / 8 : NCT % 4 == 0 ? NCT / 4 : NCT % 2 == 0 ? NCT / 2 : NCT;
}

// Workspace of the split form: [arrival counters: a region of FIXED size, so that layers of different shapes sharing one workspace never
// write partial sums over each other's counters; ZERO before the first call, left zero by every call][partial outputs]; 0 = this shape
// does not split (the workspace may then be NULL).
constexpr size_t kSplitCounterB = 64 * 1024;
extern "C" size_t se3_kpconv_fused_split_workspace_bytes(int64_t num_queries, int in_channels, int out_channels) {
  if (num_queries <= 0 || in_channels % kCC || out_channels % 32) return 0;
  const int64_t tiles = fused_tiles(num_queries);
  const int z = fused_splits(tiles, fused_colblocks(out_channels), in_channels / kCC);
  if (z == 1 || (size_t)tiles * (out_channels / 32) * sizeof(int) > kSplitCounterB) return 0;
  return kSplitCounterB + (size_t)z * tiles * kTP * kA * out_channels * sizeof(float);
}

extern "C" int se3_kpconv_so3_fused(const float* x, const void* table, int64_t num_queries, int64_t num_support, int num_neighbors,
                                    int in_channels, int out_channels, const void* weight_pieces, float* out, void* split_workspace,
                                    size_t split_workspace_bytes, int x_blocked, void* stream) {
  return se3_kpconv_so3_fused_scaled(x, table, num_queries, num_support, num_neighbors, in_channels, out_channels, weight_pieces, out,
                                     split_workspace, split_workspace_bytes, x_blocked, nullptr, stream);
}

extern "C" int se3_kpconv_so3_fused_scaled(const float* x, const void* table, int64_t num_queries, int64_t num_support, int num_neighbors,
                                           int in_channels, int out_channels, const void* weight_pieces, float* out, void* split_workspace,
                                           size_t split_workspace_bytes, int x_blocked, const float* x_amax, void* stream) {
  SE3_REQUIRE(x && table && weight_pieces && out, SE3_ERR_INVALID_ARG, "kpconv_so3_fused: null pointer");
  SE3_REQUIRE(num_neighbors >= 1 && num_neighbors <= 64, SE3_ERR_UNSUPPORTED, "kpconv_so3_fused: %d neighbours (max 64)", num_neighbors);
  SE3_REQUIRE(in_channels > 0 && in_channels % kCC == 0 && out_channels >= 32 && out_channels % 32 == 0, SE3_ERR_UNSUPPORTED,
              "kpconv_so3_fused: channels (%d, %d) must be multiples of (8, 32)", in_channels, out_channels);
  SE3_REQUIRE((int64_t)num_support * kA * in_channels < (1ll << 31), SE3_ERR_UNSUPPORTED, "kpconv_so3_fused: support features exceed 2^31 elements");
  SE3_REQUIRE(!x_blocked || in_channels % 16 == 0, SE3_ERR_UNSUPPORTED, "kpconv_so3_fused: the blocked feature layout needs in_channels %% 16 == 0 (%d)",
              in_channels);
  if (num_queries == 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  const NeighborTable t = table_views(table, num_queries, num_neighbors);
  const int NCT = out_channels / 32;
  const int64_t tiles = se3_cdiv(num_queries, kTP);
  const float* hdr = static_cast<const float*>(weight_pieces);
  const u32x4* Wf = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces) + kHeaderB);
  const size_t lds = (size_t)3 * kTileB + kSteps * 4 * sizeof(unsigned);
  int splits = 1;
  int* split_count = nullptr;
  float* split_part = nullptr;
  if (split_workspace != nullptr) {
    const size_t need = se3_kpconv_fused_split_workspace_bytes(num_queries, in_channels, out_channels);
    if (need != 0) {
      SE3_REQUIRE(split_workspace_bytes >= need, SE3_ERR_WORKSPACE, "kpconv_so3_fused: split workspace too small");
      splits = fused_splits(tiles, fused_colblocks(out_channels), in_channels / kCC);
      split_count = static_cast<int*>(split_workspace);
      split_part = reinterpret_cast<float*>(static_cast<unsigned char*>(split_workspace) + kSplitCounterB);
    }
  }
#define SE3_FUSED_X(NCW_, KS_, CT_, EXT_, BLK_)                                                                                           \
  {                                                                                                                                       \
    static bool attr_set = false;                                                                                                         \
    if (!attr_set) {                                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kpconv_fused_kernel<NCW_, KS_, CT_, EXT_, BLK_>),                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                                    \
      attr_set = true;                                                                                                                    \
    }                                                                                                                                     \
    kpconv_fused_kernel<NCW_, KS_, CT_, EXT_, BLK_>                                                                                       \
        <<<dim3((unsigned)tiles, (unsigned)(NCT / (NCW_ * CT_)), (unsigned)splits), 64 * (NCW_ * KS_ + 8), lds, st>>>(                    \
            x, t.hwt, t.nbr, t.cnt, t.NNp, Wf, hdr, num_queries, in_channels, out_channels, out, split_part, split_count, g_kpconv_variant, x_amax);                \
  }
#define SE3_FUSED(NCW_, KS_, CT_)                                  \
  {                                                                \
    if (t.NNp > 32) {                                              \
      if (x_blocked) SE3_FUSED_X(NCW_, KS_, CT_, true, true)       \
      else SE3_FUSED_X(NCW_, KS_, CT_, true, false)                \
    } else {                                                       \
      if (x_blocked) SE3_FUSED_X(NCW_, KS_, CT_, false, true)      \
      else SE3_FUSED_X(NCW_, KS_, CT_, false, false)               \
    }                                                              \
  }
  // 11 or 12 waves per compute unit (3 per SIMD: 168 registers): 8 producers + 3 or 4 consumers
  if (NCT % 8 == 0) SE3_FUSED(4, 1, 2)
  else if (NCT % 4 == 0) SE3_FUSED(4, 1, 1)
  else if (NCT % 2 == 0) SE3_FUSED(1, 3, 2)
  else SE3_FUSED(1, 3, 1)
#undef SE3_FUSED_X
#undef SE3_FUSED
  SE3_CHECK_LAUNCH("kpconv_so3_fused");
  return SE3_OK;
}

extern "C" void se3_debug_set_kpconv_variant(int variant) { g_kpconv_variant = variant; }
