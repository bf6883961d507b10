// B1: E2PN anchor-group kernel-point convolution (KPConvInterSO3), contraction stage on the f16 matrix cores at f32 accuracy.
//
// Reference: geotransformer/modules/e2pn/blocks_epn.py:454-546 (forward) with the weight permutation tables of :228-332:
//   out[p, r, d] = sum_{(s, t), c} G[p, r, (s, t), c] W[s, t, c, d],      G[p, r, (s, t), c] = H[p, orbit(s, r), anchor(t, r), c]
// with the orbit sums H of csrc/kpconv_sums.h (16 kernel-point sums x 6 anchors per point and channel, formed and split by the producer:
// csrc/kpconv_so3.hip).  Round 2's contraction rebuilt every G fragment per output anchor inside the K loop (1 or 4 LDS row reads, adds,
// a three-way bf16 split: 2.6 vector instructions per MFMA, matrix pipe 34 % busy).  Here the K loop holds NO arithmetic besides the MFMAs:
//   * operands are two f16 pieces each (x = hi + lo, 22 significant bits; the weights are scaled by a power of two so that their lo pieces
//     stay normal numbers) and the three products hi hi + hi lo + lo hi accumulate in f32: error 2^-22 per term, below the f32 GEMM's own
//     accumulation error (tests/test_gpu_ops.py::test_kpconv_matrix_core_path_has_f32_accuracy) -- 3 MFMAs where the bf16 form took 6;
//   * a workgroup owns a 16-point tile = 96 output rows = three 32-row tiles (anchor pairs) of v_mfma_f32_32x32x16_f16 and, per wave, one
//     32-column tile: a wave's K16-step (2 weight slots x 8 channels) is 6 ds_read_b128 (A: the H tile image in LDS, read in place through
//     a 72-entry offset table), 2 global loads (B: weight fragments in lane order, requested a step ahead, L1 / L2 resident) and 9 MFMAs;
//   * the row order inside a 32-row tile is chosen so that every 16-lane group of a ds_read_b128 reads ONE run of 16 different points:
//     conflict-free with the odd row stride of the image.
// Layers with fewer than 4 column tiles split the K16-steps of a chunk over the waves instead (sums merged through LDS at the end).
#include "common.h"
#include "kpconv_sums.h"

namespace {

using namespace kpsum;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;      // (an array of HIP uint4 structs ends up in scratch)

#ifdef SE3_KPCONV_STAMPS      // diagnostic build of tools/micro/kpconv_stamps.hip only: wave time stamps of the first workgroups
__device__ long long* g_stamps = nullptr;
constexpr int kStampBlocks = 64, kStampSteps = 40, kStampSlots = 6;
#define SE3_STAMP(step_, slot_)                                                                                            \
  if (g_stamps && blockIdx.x < kStampBlocks && blockIdx.y == 0 && (step_) < kStampSteps && lane == 0)                      \
    g_stamps[(((int64_t)blockIdx.x * 16 + wave) * kStampSteps + (step_)) * kStampSlots + (slot_)] = __builtin_amdgcn_s_memtime();
#else
#define SE3_STAMP(step_, slot_)
#endif

constexpr int kTile4 = kTileB / 16;          // uint4 per tile image (3104)
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
  const int a_base = row_point(i32) * kRowB;                           // + piece * kTP * kRowB + run * 16
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
        a[rt][1] = *reinterpret_cast<const f16x8*>(lds + off + kTP * kRowB);
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


// ======== fused form: the orbit sums never leave the compute unit ========================================================================
// One 16-wave workgroup per compute unit and 16-point tile; the tile image exists twice in LDS (2 x 48.5 KB).  In phase t the PRODUCER
// waves form the image of channel chunk t while the CONSUMER waves multiply chunk t - 1 out of the other buffer; one barrier per chunk.
//   consumer waves (NCW x KS): the K loop -- ds_read_b128 of the image, weight fragments from L1 / L2 (a ring of three K16-steps in
//     registers), MFMAs; no other arithmetic;
//   producer waves (NPW): one POINT of the tile at a time, lane = (anchor a, channel c of the chunk) (48 of 64 lanes); the point's neighbour
//     list and its 15 influence weights per neighbour come from a per-layer table (kpconv_neighbor_table_kernel) through SCALAR loads --
//     they are uniform over the wave, so the inner loop is 1 vector load + 8 packed FMAs with SGPR-pair operands per neighbour, no LDS;
//     then the 16 orbit sums, the f16 hi / lo split and a DPP exchange inside channel pairs so that every lane stores one dword per orbit.
// The matrix pipe (consumers) and the vector ALUs (producers) of a SIMD are fed by different waves and overlap.

// neighbour table of a layer: per query point the VALID neighbours compacted to the front (invalid ones carry weight 0 in the reference:
// blocks_epn.py:471,377 shadow point / zero feature row), padded with (index 0, weights 0) up to a multiple of 24 (NNp)
//   nbr  [P][NNp] int32     support row of the j-th valid neighbour
//   wts  [P][NNp][16] f32   w[k] = max(0, 1 - |s - q - kp_k| / sigma), k < 15; [15] = 0
//   cnt  [P] int32          valid neighbours
__global__ __launch_bounds__(64) void kpconv_neighbor_table_kernel(const float* __restrict__ q_pts, const float* __restrict__ s_pts,
                                                                   const int64_t* __restrict__ idx, int64_t Ns, int NN, int NNp,
                                                                   const float* __restrict__ kp, float inv_sigma, int* __restrict__ nbr,
                                                                   float* __restrict__ wts, int* __restrict__ cnt) {
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
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = 0.f;
    int row = 0;
    if (e < nv) {
      const int64_t js = idx[p * NN + order[e]];
      row = (int)js;
      const float qx = q_pts[3 * p], qy = q_pts[3 * p + 1], qz = q_pts[3 * p + 2];
      const float sx = s_pts[3 * js], sy = s_pts[3 * js + 1], sz = s_pts[3 * js + 2];
#pragma unroll
      for (int k = 0; k < kK; k++) {
        const float dx = sx - qx - kp[3 * k], dy = sy - qy - kp[3 * k + 1], dz = sz - qz - kp[3 * k + 2];
        w[k] = fmaxf(0.f, 1.f - sqrtf(dx * dx + dy * dy + dz * dz) * inv_sigma);
      }
    }
    nbr[p * NNp + e] = row;
    float4* dst = reinterpret_cast<float4*>(wts + (p * NNp + e) * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) dst[q] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  }
}

template <int NCW, int KS, int CT>      // consumer waves: NCW column groups x KS K-split groups; CT column tiles (32 columns) per wave
__global__ __launch_bounds__(64 * (NCW * KS + 8)) void kpconv_fused_kernel(
    const float* __restrict__ x, const int* __restrict__ nbr, const float* __restrict__ wts, const int* __restrict__ cnt, int NNp,
    const u32x4* __restrict__ Wf, const float* __restrict__ hdr, int64_t P, int Cin, int Cout, float* __restrict__ out) {
  constexpr int NC = NCW * KS;                                         // consumer waves
  constexpr int NPW = 8;                                               // producer waves: two points of the tile each
  constexpr int kSPW = kSteps / KS;                                    // K16-steps per consumer wave and chunk
  static_assert(kSteps % KS == 0 && kSPW % 2 == 0, "K split must leave an even number of K16-steps per wave");
  extern __shared__ __align__(16) unsigned char lds[];                 // [3 images][table]
  unsigned* tab = reinterpret_cast<unsigned*>(lds + 3 * kTileB);       // [K16-step][rsel][h]: three 8-bit run numbers (anchor pairs 0..2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x, p0 = tile * kTP;
  const int chunks = Cin / kCC, pairs = (chunks + 1) / 2;
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
  // Schedule: step u = 0 .. chunks + 1, one barrier between steps.  Consumers multiply chunk u - 2 (image (u - 2) % 3) in step u >= 2.
  // A producer wave handles ONE point per step over a PAIR of chunks (16 channels: lane = anchor x channel pair, 8-byte loads, whole
  // 64-byte sectors): in step u = 2T its first point, in step 2T + 1 its second, for chunks (2T, 2T + 1).  Only one image is free for
  // writing besides the one being filled while three exist, so rows of chunk 2T go to image (2T) % 3 at once, while rows of chunk 2T + 1
  // computed in step 2T wait in registers for one step (image (2T + 1) % 3 is still being read during step 2T).
  const int steps_total = chunks + 2;
  if (wave >= NC) {
    // ---------------- producer ----------------
    // The gather is bound by the bytes in flight per compute unit (random 64-byte sectors out of L2 / Infinity Cache): 24 neighbour rows
    // are requested per wave before the first is used (8 waves x 24 x 384 B = 72 KB in flight).
    const int pw = wave - NC;
    const int a = lane < 48 ? lane >> 3 : 5, cp = lane & 7, half = cp >> 2;    // channel pair cp of the 16: chunk half, channels 2 (cp & 3) + {0, 1}
    const unsigned col0 = (unsigned)(a * Cin + 2 * cp);                  // + pair * 16
    const unsigned rowlen = (unsigned)(kA * Cin);
    const int dst0 = a * 16 + (cp & 3) * 4;                               // byte offset of the lane's hi dword inside a point's row (+ o * 96; lo: + kTP * kRowB)
    unsigned held[2 * kOrbits];
#pragma unroll
    for (int o = 0; o < 2 * kOrbits; o++) held[o] = 0u;
    for (int u = 0; u < steps_total; u++) {
      SE3_STAMP(u, 0)
      if (u < 2 * pairs) {
        const int T = u >> 1, second = u & 1;
        const int i = pw + NPW * second;                                  // point of the tile, uniform over the wave
        const int64_t p = p0 + i;
        const bool has2 = 2 * T + 1 < chunks;                             // (odd chunk count: the last pair is a single chunk)
        unsigned char* img_lo = lds + ((2 * T) % 3) * kTileB;
        unsigned char* img_hi = lds + ((2 * T + 1) % 3) * kTileB;
        const bool active = lane < 48 && (half == 0 || has2);
        if (second && half == 1 && active) {                              // the first point's rows of chunk 2T + 1, held since the last step
          unsigned char* dst = img_hi + pw * kRowB + dst0;
#pragma unroll
          for (int o = 0; o < kOrbits; o++) {
            *reinterpret_cast<unsigned*>(dst + o * (kA * 16)) = held[2 * o];
            *reinterpret_cast<unsigned*>(dst + o * (kA * 16) + kTP * kRowB) = held[2 * o + 1];
          }
        }
        float f0[kK], f1[kK];
#pragma unroll
        for (int k = 0; k < kK; k++) f0[k] = f1[k] = 0.f;
        if (p < P) {
          const int nv = cnt[p];
          const int* nb = nbr + p * NNp;
          const float* wr = wts + p * NNp * 16;
          const float* xc = x + T * 16 + (active ? col0 : 0u);
          // rotating prefetch, three batches of 8 neighbour rows deep: the requests of batch b + 3 leave right behind the FMAs of batch b.
          // All requests are unconditional (batch numbers clamped into the table row, whose tail is index 0 / weight 0) and at the top
          // level of the loop body: hipcc sinks a load into the branch that uses it, which would serialise request and use.
          const int nbat = (nv + 7) >> 3, lastb = NNp / 8 - 1;
          SE3_STAMP(u, 1)
          float2 xa[8], xb[8], xc8[8];
#define SE3_REQ(dst_, b_)                                                                                     \
  {                                                                                                           \
    const int bb_ = (b_) < lastb ? (b_) : lastb;                                                              \
    _Pragma("unroll") for (int q = 0; q < 8; q++) dst_[q] = *reinterpret_cast<const float2*>(xc + (unsigned)nb[bb_ * 8 + q] * rowlen); \
  }
#define SE3_FMA(src_, b_)                                                                                     \
  _Pragma("unroll") for (int q = 0; q < 8; q++) _Pragma("unroll") for (int k = 0; k < kK; k++) {              \
    const float w = wr[((b_) * 8 + q) * 16 + k];                                                              \
    f0[k] = fmaf(w, src_[q].x, f0[k]);                                                                        \
    f1[k] = fmaf(w, src_[q].y, f1[k]);                                                                        \
  }
          SE3_REQ(xa, 0) SE3_REQ(xb, 1) SE3_REQ(xc8, 2)
          for (int b = 0; b < nbat; b += 3) {
            SE3_FMA(xa, b)
            SE3_REQ(xa, b + 3)
            if (b + 1 < nbat) { SE3_FMA(xb, b + 1) }
            SE3_REQ(xb, b + 4)
            if (b + 2 < nbat) { SE3_FMA(xc8, b + 2) }
            SE3_REQ(xc8, b + 5)
          }
#undef SE3_REQ
#undef SE3_FMA
        }
        SE3_STAMP(u, 2)
        unsigned char* dst = (half ? img_hi : img_lo) + i * kRowB + dst0;
        const bool store_now = active && (half == 0 || second);
#pragma unroll
        for (int o = 0; o < kOrbits; o++) {
          float v0 = 0.f, v1 = 0.f;
#pragma unroll
          for (int k = 0; k < kK; k++)
            if ((kOrb.mask[o] >> k) & 1) {
              v0 += f0[k];
              v1 += f1[k];
            }
          const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
          const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
          const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
          const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
          if (store_now) {
            *reinterpret_cast<unsigned*>(dst + o * (kA * 16)) = hi;
            *reinterpret_cast<unsigned*>(dst + o * (kA * 16) + kTP * kRowB) = lo;
          }
          held[2 * o] = hi;
          held[2 * o + 1] = lo;
        }
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
  const int a_base = row_point(i32) * kRowB;                           // + piece * kTP * kRowB + run * 16
  const int tab_lane = row_rsel(i32) * 2 + h;
  f32x16 acc[CT][3];
#pragma unroll
  for (int n = 0; n < CT; n++)
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[n][rt][v] = 0.f;
  // weight fragments (hi, lo) of this wave's next two K16-steps: two register sets, each refilled behind the MFMAs that read it
  const u32x4* wbase = Wf + (int64_t)ct0 * 2 * 64 + lane;
  const int64_t wstep = (int64_t)NCT * 2 * 64;                          // uint4 per K16-step over the whole layer
  const int64_t last_step = (int64_t)chunks * kSteps - KS + ksp;
  u32x4 bq[2][CT][2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    int64_t g = ksp + j * KS;
    g = g < last_step ? g : last_step;
#pragma unroll
    for (int n = 0; n < CT; n++) {
      bq[j][n][0] = wbase[g * wstep + n * 128];
      bq[j][n][1] = wbase[g * wstep + n * 128 + 64];
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
        av[0][rt][1] = *reinterpret_cast<const f16x8*>(img + off + kTP * kRowB);
      }
    }
#pragma unroll 1
    for (int q2 = 0; q2 < kSPW; q2 += 2) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int st = ksp + (q2 + j) * KS;
        {
          const int sn = st + KS < kSteps ? st + KS : st;                // next step of this chunk (clamped: the last one re-reads itself)
          const unsigned runs = tab[sn * 4 + tab_lane];
#pragma unroll
          for (int rt = 0; rt < 3; rt++) {
            const int off = a_base + (int)((runs >> (8 * rt)) & 0xff) * 16;
            av[j ^ 1][rt][0] = *reinterpret_cast<const f16x8*>(img + off);
            av[j ^ 1][rt][1] = *reinterpret_cast<const f16x8*>(img + off + kTP * kRowB);
          }
        }
        // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b0 = __builtin_bit_cast(f16x8, bq[j][n][0]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[j][rt][1], b0, acc[n][rt], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b1 = __builtin_bit_cast(f16x8, bq[j][n][1]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[j][rt][0], b1, acc[n][rt], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b0 = __builtin_bit_cast(f16x8, bq[j][n][0]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[j][rt][0], b0, acc[n][rt], 0, 0, 0);
        }
        {
          int64_t g = (int64_t)cc * kSteps + st + 2 * KS;                // unconditional (clamped) so that the compiler can count the requests
          g = g < last_step ? g : last_step;
#pragma unroll
          for (int n = 0; n < CT; n++) {
            bq[j][n][0] = wbase[g * wstep + n * 128];
            bq[j][n][1] = wbase[g * wstep + n * 128 + 64];
          }
        }
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
  const float inv_scale = hdr[0];
#pragma unroll
  for (int n = 0; n < CT; n++)
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int64_t p = p0 + v;
        const int r = 2 * rt + ((0x96 >> (2 * (v >> 2) + h)) & 1);
        if (p < P) out[(p * kA + r) * Cout + (ct0 + n) * 32 + i32] = acc[n][rt][v] * inv_scale;
      }
}

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
  const size_t nnp = (size_t)(num_neighbors + 23) / 24 * 24;
  return (size_t)num_queries * nnp * (16 * sizeof(float) + sizeof(int)) + (size_t)num_queries * sizeof(int) + 256;
}

// One call = KPConvInterSO3.forward (blocks_epn.py:454-546): neighbour table, then the fused gather + contraction.  `workspace`:
// se3_kpconv_neighbor_table_bytes bytes; weight_pieces from se3_kpconv_split_weights_f16.
extern "C" int se3_kpconv_so3_fused(const float* q_pts, const float* s_pts, const int64_t* idx, const float* x,
                                    const float* kernel_points_dev, float sigma, int64_t num_queries, int64_t num_support,
                                    int num_neighbors, int in_channels, int out_channels, const void* weight_pieces, float* out,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(q_pts && s_pts && idx && x && kernel_points_dev && weight_pieces && out && workspace, SE3_ERR_INVALID_ARG,
              "kpconv_so3_fused: null pointer");
  SE3_REQUIRE(num_neighbors >= 1 && num_neighbors <= 64, SE3_ERR_UNSUPPORTED, "kpconv_so3_fused: %d neighbours (max 64)", num_neighbors);
  SE3_REQUIRE(in_channels > 0 && in_channels % kCC == 0 && out_channels >= 32 && out_channels % 32 == 0 && sigma > 0.f, SE3_ERR_UNSUPPORTED,
              "kpconv_so3_fused: channels (%d, %d) must be multiples of (8, 32)", in_channels, out_channels);
  SE3_REQUIRE((int64_t)num_support * kA * in_channels < (1ll << 31), SE3_ERR_UNSUPPORTED, "kpconv_so3_fused: support features exceed 2^31 elements");
  SE3_REQUIRE(workspace_bytes >= se3_kpconv_neighbor_table_bytes(num_queries, num_neighbors), SE3_ERR_INVALID_ARG,
              "kpconv_so3_fused: workspace too small");
  if (num_queries == 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  const int NNp = (num_neighbors + 23) / 24 * 24;
  float* wts = static_cast<float*>(workspace);
  int* nbr = reinterpret_cast<int*>(wts + (size_t)num_queries * NNp * 16);
  int* cnt = nbr + (size_t)num_queries * NNp;
  kpconv_neighbor_table_kernel<<<(unsigned)num_queries, 64, 0, st>>>(q_pts, s_pts, idx, num_support, num_neighbors, NNp, kernel_points_dev,
                                                                    1.0f / sigma, nbr, wts, cnt);
  const int NCT = out_channels / 32;
  const int64_t tiles = se3_cdiv(num_queries, kTP);
  const float* hdr = static_cast<const float*>(weight_pieces);
  const u32x4* Wf = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces) + kHeaderB);
  const size_t lds = (size_t)3 * kTileB + kSteps * 4 * sizeof(unsigned);
#define SE3_FUSED(NCW_, KS_, CT_)                                                                                                   \
  {                                                                                                                                 \
    static bool attr_set = false;                                                                                                   \
    if (!attr_set) {                                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kpconv_fused_kernel<NCW_, KS_, CT_>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                              \
      attr_set = true;                                                                                                              \
    }                                                                                                                               \
    kpconv_fused_kernel<NCW_, KS_, CT_><<<dim3((unsigned)tiles, (unsigned)(NCT / (NCW_ * CT_))), 64 * (NCW_ * KS_ + 8), lds, st>>>( \
        x, nbr, wts, cnt, NNp, Wf, hdr, num_queries, in_channels, out_channels, out);                                               \
  }
  // 12 waves per compute unit (3 per SIMD: 168 registers): 8 producers + 3 or 4 consumers
  if (NCT % 8 == 0) SE3_FUSED(4, 1, 2)
  else if (NCT % 4 == 0) SE3_FUSED(4, 1, 1)
  else if (NCT % 2 == 0) SE3_FUSED(1, 3, 2)
  else SE3_FUSED(1, 3, 1)
#undef SE3_FUSED
  SE3_CHECK_LAUNCH("kpconv_so3_fused");
  return SE3_OK;
}
