// B1: E2PN anchor-group kernel-point convolution (KPConvInterSO3), contraction stage on the matrix cores.
//
// Reference: geotransformer/modules/e2pn/blocks_epn.py:454-546 (forward) with the weight permutation tables of :228-332:
//   out[p, r, d] = sum_{k, a, c} F[p, k, a, c] W[kidx[k, r], ridx[a, r], c, d],   F[p, k, a, c] = sum_n w[p, n, k] x[idx[p, n], a, c]
// (F = the kernel-point sums of csrc/kpconv_so3.hip).  Summing the (k, a) slices that share a weight slot under output anchor r first,
//   G[p, r, (s, t), c] = sum_{k: kidx[k, r] = s} F[p, k, a: ridx[a, r] = t, c],        out[(p, r), d] = G[(p, r), (s, t, c)] . W[(s, t, c), d]
// is one GEMM with 2.5x fewer flops than the reference's expanded form.  Round 1 wrote G (P*6, 36 Cin) to HBM (1.2 - 2.9 GB per call)
// and multiplied it with a library GEMM at the f32 MFMA rate.  This kernel never forms G in memory:
//   * the gather kernel leaves F (2.4x smaller than G) in tile order [channel chunk][point][732] (15 x 6 x 8 values, zero slot, pad);
//   * a workgroup owns 16 points (= six 16-row MFMA tiles, one per output anchor r) and a block of up to 64 output columns and streams the
//     channel chunks; per K-step of 4 weight slots x 8 channels every wave builds the G fragment(s) of its output anchor(s) from the F
//     tile in LDS (1 or 4 adds per element: the C4 orbits of the kernel points) directly in MFMA operand order, in registers;
//   * the product runs on the bf16 matrix cores at f32 accuracy: every f32 operand is split into three bf16 pieces
//     (a = a1 + a2 + a3 exactly: 3 x 8 significant bits), the weights once per call (se3_kpconv_split_weights), G on the fly, and the six
//     products a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1 (everything above 2^-24 relative) accumulate in f32:
//     6 x v_mfma_f32_16x16x32_bf16 = 96 cycles for what 8 x v_mfma_f32_16x16x4_f32 do in 256.
// The weight fragments are streamed from L2 / L1 in lane order (1 KB per fragment, pre-arranged by the split kernel).  Layers wider than 64
// columns run one workgroup per (point tile, block of 64 columns): the fragments are rebuilt per block (a fifth of the time), nothing is
// shared through LDS and no barrier sits inside a channel chunk.  (The round also built a dedicated wide kernel -- 8 waves in two role
// groups sharing the G fragments through LDS, LDS-DMA tile copies, raw s_barrier: 1.6-1.76 ms on the 128 / 256-column layers against
// 1.2-1.3 ms for the column-block form here; DESIGN.md section 4 keeps its measurements.)
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kK = 15, kA = 6, kS = 6;
constexpr int kTP = 16;                       // points per workgroup tile
constexpr int kCC = 8;                        // channels per chunk
constexpr int kSlots = kS * kA;               // 36 weight slots (s, t)
constexpr int kKS = kSlots / 4;               // K-steps per channel chunk: 4 slots x 8 channels = 32
constexpr int kRow = kK * kA * kCC;           // 720 floats of F per point and chunk
// Row stride per point, in global memory AND in LDS (a tile is one linear 46.8 KB copy): 720 values + an 8-float zero slot (absent orbit
// members of single-member slots read it) + 4 pad = 732 floats = 2928 B = 16 B x 183 (odd: conflict-free b128 reads over 16 points).
constexpr int kRowPad = 732;
constexpr int kZeroSlot = kRow;               // float offset of the zero slot inside a row
constexpr int kTileFloats = kTP * kRowPad;    // 11712 floats = 2928 float4

// slot tables of the SE3ET configuration (se3et_amd/tables.py; identical to csrc/kpconv_so3.hip)
__device__ constexpr int kKidx[kK][kA] = {{0, 1, 1, 1, 1, 2}, {1, 0, 1, 2, 1, 1}, {1, 1, 0, 1, 2, 1}, {1, 2, 1, 0, 1, 1},
                                          {1, 1, 2, 1, 0, 1}, {2, 1, 1, 1, 1, 0}, {3, 3, 3, 4, 4, 4}, {3, 4, 3, 3, 4, 4},
                                          {3, 4, 4, 3, 3, 4}, {3, 3, 4, 4, 3, 4}, {4, 3, 3, 4, 4, 3}, {4, 4, 3, 3, 4, 3},
                                          {4, 4, 4, 3, 3, 3}, {4, 3, 4, 4, 3, 3}, {5, 5, 5, 5, 5, 5}};
__device__ constexpr int kRidx[kA][kA] = {{0, 3, 3, 3, 3, 5}, {1, 0, 4, 5, 2, 1}, {2, 2, 0, 4, 5, 4},
                                          {3, 5, 2, 0, 4, 3}, {4, 4, 5, 2, 0, 2}, {5, 1, 1, 1, 1, 0}};
// K order of the slots: the 18 slots whose kernel-point orbit has ONE member (s in {0, 2, 5}) first, then the 18 with FOUR (s in {1, 3, 4}):
// K-steps 0..3 read one F row per element, 5..8 four, step 4 is mixed (absent members read the zero block).
__host__ __device__ constexpr int slot_at(int pos) {
  const int single[3] = {0, 2, 5}, quad[3] = {1, 3, 4};
  return pos < 18 ? single[pos / 6] * 6 + pos % 6 : quad[(pos - 18) / 6] * 6 + (pos - 18) % 6;
}

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {      // round-to-nearest-even pair -> one dword (v_cvt_pk_bf16_f32)
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// a[0..7] (f32) -> three bf16x8 fragments with a = p1 + p2 + p3 (to 2^-25 |a|)
__device__ __forceinline__ void split3(const float (&a)[8], uint4& p1, uint4& p2, uint4& p3) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float x0 = a[2 * i], x1 = a[2 * i + 1];
    h[i] = pack_bf16(x0, x1);
    const float r0 = x0 - bf16_lo(h[i]), r1 = x1 - bf16_hi(h[i]);
    m[i] = pack_bf16(r0, r1);
    const float q0 = r0 - bf16_lo(m[i]), q1 = r1 - bf16_hi(m[i]);
    l[i] = pack_bf16(q0, q1);
  }
  p1 = make_uint4(h[0], h[1], h[2], h[3]);
  p2 = make_uint4(m[0], m[1], m[2], m[3]);
  p3 = make_uint4(l[0], l[1], l[2], l[3]);
}

// ---- weights: (36 Cin, Cout) f32, row (s*6 + t) * Cin + c  ->  bf16 fragments [K-step][column tile][piece][lane][8] ---------------------
__global__ void kpconv_split_weights_kernel(const float* __restrict__ W, int Cin, int Cout, uint4* __restrict__ Wf) {
  const int NT = Cout / 16;
  const int64_t frag = blockIdx.x;                    // (K-step, column tile)
  const int ksg = (int)(frag / NT), nt = (int)(frag % NT);
  const int cc = ksg / kKS, ks = ksg % kKS;
  const int lane = threadIdx.x, kb = lane >> 4, n = nt * 16 + (lane & 15);
  const int st = slot_at(4 * ks + kb);
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = W[((int64_t)st * Cin + cc * kCC + j) * Cout + n];
  uint4 p1, p2, p3;
  split3(w, p1, p2, p3);
  uint4* dst = Wf + frag * 3 * 64 + lane;
  dst[0] = p1;
  dst[64] = p2;
  dst[128] = p3;
}

// ---- contraction -------------------------------------------------------------------------------------------------------------------
struct SlotEntry { unsigned short off[4]; };           // float offsets (k*6 + a) * 8 of the orbit members inside a point's F row

// ---- one wave per output anchor (or two), fragments in registers ---------------------------------------------
// With few output columns a K-step holds little matrix work per G fragment (6 MFMAs per column tile) and the kernel is bound by BUILDING
// the fragments.  So the six row tiles (output anchors) go to six waves: wave r builds its own fragment of the K-step in registers and
// multiplies it with all NT column tiles -- no fragment buffer in LDS, no barrier inside a channel chunk (the round's first narrow kernel
// shared the G fragments through LDS behind three barriers per K-step: 1.6-1.8x slower).  Every wave reads the K-step's weight fragments
// (NT x 3 KB, requested one K-step ahead) straight from global memory: the six waves of the three resident workgroups hit the same lines
// in L1.  (Fetching them once per workgroup into a double-buffered LDS tile behind one barrier per K-step measured 1-15 % SLOWER: the
// 12 KB per wave and K-step then come out of LDS, which the fragment builder is already loading.  Building the fragments of K-step ks + 1
// in the shadow of the MFMAs of K-step ks inside one wave -- the K-steps of a chunk unrolled, `sched_group_barrier` groups of one MFMA and
// four VALU instructions: the interleaving comes out as asked, at 200 VGPRs, and the layer takes 1.63 ms instead of 1.33.)
__device__ __forceinline__ void build_fragment(const float* frow, const SlotEntry en, bool multi, bf16x8& a1, bf16x8& a2, bf16x8& a3) {
  float v[8];
  {
    const float4 x0 = *reinterpret_cast<const float4*>(frow + en.off[0]);
    const float4 x1 = *reinterpret_cast<const float4*>(frow + en.off[0] + 4);
    v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
  }
  if (multi) {                                                          // block-uniform: K-steps 0..3 hold single-member slots only
#pragma unroll
    for (int j = 1; j < 4; j++) {
      const float* sp = frow + en.off[j];
      const float4 x0 = *reinterpret_cast<const float4*>(sp);
      const float4 x1 = *reinterpret_cast<const float4*>(sp + 4);
      v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
    }
  }
  uint4 p1, p2, p3;
  split3(v, p1, p2, p3);
  a1 = __builtin_bit_cast(bf16x8, p1); a2 = __builtin_bit_cast(bf16x8, p2); a3 = __builtin_bit_cast(bf16x8, p3);
}

template <int NT, int RW, int DBG = 0>       // DBG: ablation bits for tools/micro (1 no weight loads, 2 no builds, 4 no MFMAs, 8 no F tile copies); column tiles (16 output channels each), 1..4; output anchors per wave (1: 6 waves, 2: 3 waves)
__global__ __launch_bounds__(384 / RW) __attribute__((amdgpu_waves_per_eu(RW == 1 ? 3 : 2, RW == 1 ? 4 : 3))) void kpconv_contract_rows_kernel(
    const float* __restrict__ F, const uint4* __restrict__ Wf, int64_t P, int64_t P16, int Cin, int Cout, float* __restrict__ out) {
  extern __shared__ __align__(16) float lds[];
  float* ftile = lds;                                                   // [16 points][732]
  SlotEntry* tab = reinterpret_cast<SlotEntry*>(lds + kTileFloats);     // [9 K-steps][6 r][4 kb]
  const int tid = threadIdx.x, lane = tid & 63, r0 = (tid >> 6) * RW;   // wave = RW output anchors
  const int64_t p0 = (int64_t)blockIdx.x * kTP;
  for (int e = tid; e < kKS * kA * 4; e += 384 / RW) {
    const int kbq = e & 3, rr = (e >> 2) % kA, ks = e / (4 * kA);
    const int st = slot_at(4 * ks + kbq), sl = st / kA, t = st % kA;
    int a = 0;
    for (int aa = 0; aa < kA; aa++) a = kRidx[aa][rr] == t ? aa : a;
    unsigned long long packed = (unsigned long long)kZeroSlot * 0x0001000100010001ull;   // four 16-bit offsets; absent member = zero slot
    int cnt = 0;
    for (int k = 0; k < kK; k++)
      if (kKidx[k][rr] == sl && cnt < 4) {
        const unsigned long long off = (unsigned long long)((k * kA + a) * kCC);
        packed = (packed & ~(0xffffull << (16 * cnt))) | (off << (16 * cnt));
        cnt++;
      }
    reinterpret_cast<unsigned long long*>(tab)[e] = packed;
  }
  f32x4 acc[RW][NT];
#pragma unroll
  for (int q = 0; q < RW; q++)
#pragma unroll
    for (int n = 0; n < NT; n++) acc[q][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int chunks = Cin / kCC;
  const int64_t steps = (int64_t)chunks * kKS;
  const int prow = lane & 15, kb = lane >> 4;
  const float* frow = ftile + prow * kRowPad;
  constexpr int kFr = 3 * NT;                         // weight fragments of a K-step (a fragment = 64 lanes x 16 B, lane-linear)
  // NT <= 4: the next K-step's fragments are in flight while this step is built and multiplied (double buffered in registers); NT = 8
  // (96 registers of fragments, 64 of accumulators): this step's fragments are requested at the top of the step, behind the build
  constexpr bool kSingle = NT > 4;
  uint4 bn[kSingle ? 1 : kFr];
  // blockIdx.y: the workgroup's block of NT column tiles (layers wider than 64 columns run one workgroup per (point tile, column block):
  // every block rebuilds the fragments -- a fifth of the time -- and nothing is shared through LDS)
  const int nt0 = blockIdx.y * NT;
  const int64_t step_frags = (int64_t)(Cout / 16) * 3;                  // fragments of a K-step over the whole layer
  Wf += (int64_t)nt0 * 3 * 64;
  if (!kSingle) {
#pragma unroll
    for (int f = 0; f < kFr; f++) bn[kSingle ? 0 : f] = Wf[f * 64 + lane];
  }
  int64_t g = 0;
  // The F tile of chunk cc + 1 (46.8 KB) is requested into registers at the top of chunk cc and stored to LDS at the chunk boundary: the
  // plain copy loop (load, store, load, ...) paid one HBM round trip per float4 and thread -- two thirds of the kernel time.
  constexpr int kThreads = 384 / RW, kPF = (kTileFloats / 4 + kThreads - 1) / kThreads;
  f32x4 pf[kSingle ? 1 : kPF];                        // (ext_vector_type: an array of HIP float4 structs ends up in scratch)
  if (!kSingle) {
    const f32x4* src = reinterpret_cast<const f32x4*>(F + p0 * kRowPad);
#pragma unroll
    for (int i = 0; i < kPF; i++) pf[kSingle ? 0 : i] = src[min(tid + i * kThreads, kTileFloats / 4 - 1)];
  }
  for (int cc = 0; cc < chunks; cc++) {
    __syncthreads();                                                    // the previous chunk's F tile is no longer read
    if (!kSingle) {
#pragma unroll
      for (int i = 0; i < kPF; i++)
        if (tid + i * kThreads < kTileFloats / 4) reinterpret_cast<f32x4*>(ftile)[tid + i * kThreads] = pf[kSingle ? 0 : i];
      if (!(DBG & 8) || cc == 0) {
        const int cn = cc + 1 < chunks ? cc + 1 : cc;                    // unconditional (clamped) so that the compiler can count the requests
        const f32x4* src = reinterpret_cast<const f32x4*>(F + ((int64_t)cn * P16 + p0) * kRowPad);
#pragma unroll
        for (int i = 0; i < kPF; i++) pf[kSingle ? 0 : i] = src[min(tid + i * kThreads, kTileFloats / 4 - 1)];
      }
    } else {                                                            // no registers to spare for the tile: plain batched copy
      const f32x4* src = reinterpret_cast<const f32x4*>(F + ((int64_t)cc * P16 + p0) * kRowPad);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        f32x4 t[kPF / 2];
#pragma unroll
        for (int i = 0; i < kPF / 2; i++) t[i] = src[min(tid + (h * (kPF / 2) + i) * kThreads, kTileFloats / 4 - 1)];
#pragma unroll
        for (int i = 0; i < kPF / 2; i++)
          if (tid + (h * (kPF / 2) + i) * kThreads < kTileFloats / 4) reinterpret_cast<f32x4*>(ftile)[tid + (h * (kPF / 2) + i) * kThreads] = t[i];
      }
    }
    __syncthreads();
#define SE3_PRODUCT(a_, pc_)                                                                                        \
  _Pragma("unroll") for (int q = 0; q < RW; q++) _Pragma("unroll") for (int n = 0; n < NT; n++)                  \
      acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_[q], b[n][pc_], acc[q][n], 0, 0, 0);
    bf16x8 a1[RW], a2[RW], a3[RW];
    if (DBG & 2) {
#pragma unroll
      for (int q = 0; q < RW; q++) build_fragment(frow, tab[(r0 + q) * 4 + kb], false, a1[q], a2[q], a3[q]);
    }
#pragma unroll 1
    for (int ks = 0; ks < kKS; ks++, g++) {
      bf16x8 b[NT][3];
      if (kSingle) {
        const uint4* src = Wf + g * step_frags * 64 + lane;
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
          for (int pc = 0; pc < 3; pc++) b[n][pc] = __builtin_bit_cast(bf16x8, src[(n * 3 + pc) * 64]);
      }
      if (!(DBG & 2)) {
#pragma unroll
        for (int q = 0; q < RW; q++) build_fragment(frow, tab[(ks * kA + r0 + q) * 4 + kb], ks >= 4, a1[q], a2[q], a3[q]);
      }
      if (!kSingle) {
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
          for (int pc = 0; pc < 3; pc++) b[n][pc] = __builtin_bit_cast(bf16x8, bn[kSingle ? 0 : n * 3 + pc]);
        if (!(DBG & 1)) {                                               // request step g + 1 (clamped: unconditional, so the compiler counts it)
          const int64_t gq = g + 1 < steps ? g + 1 : steps - 1;
          const uint4* src = Wf + gq * step_frags * 64 + lane;
#pragma unroll
          for (int f = 0; f < kFr; f++) bn[kSingle ? 0 : f] = src[f * 64];
        }
      }
      // product-major over the tiles: consecutive MFMAs go to different accumulators; smallest terms first
      if (!(DBG & 4)) {
        SE3_PRODUCT(a3, 0)
        SE3_PRODUCT(a1, 2)
        SE3_PRODUCT(a2, 1)
        SE3_PRODUCT(a2, 0)
        SE3_PRODUCT(a1, 1)
        SE3_PRODUCT(a1, 0)
      } else {
#pragma unroll
        for (int q = 0; q < RW; q++)
#pragma unroll
          for (int n = 0; n < NT; n++) {
            const bf16x8 t = a1[q] + a2[q] + a3[q] + b[n][0] + b[n][1] + b[n][2];
            acc[q][n][0] += (float)t[0];
          }
      }
    }
#undef SE3_PRODUCT
  }
  // accumulator tile: lane holds column (lane & 15), rows (lane >> 4) * 4 + i = points of the tile
#pragma unroll
  for (int q = 0; q < RW; q++)
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int64_t p = p0 + (lane >> 4) * 4 + i;
        if (p < P) out[(p * kA + r0 + q) * Cout + (nt0 + n) * 16 + (lane & 15)] = acc[q][n][i];
      }
}

}  // namespace

extern "C" size_t se3_kpconv_points_floats(int64_t num_queries, int in_channels) {
  if (num_queries < 0 || in_channels <= 0 || in_channels % kCC) return 0;
  return (size_t)(in_channels / kCC) * (size_t)(se3_cdiv(num_queries, kTP) * kTP) * kRowPad + 64;      // + one partial DMA piece
}

extern "C" size_t se3_kpconv_weight_fragments_bytes(int in_channels, int out_channels) {
  if (in_channels <= 0 || out_channels <= 0 || in_channels % kCC || out_channels % 16) return 0;
  return (size_t)(in_channels / kCC) * kKS * (out_channels / 16) * 3 * 64 * sizeof(uint4);
}

extern "C" int se3_kpconv_split_weights(const float* weights, int in_channels, int out_channels, void* fragments, void* stream) {
  SE3_REQUIRE(weights && fragments, SE3_ERR_INVALID_ARG, "kpconv_split_weights: null pointer");
  SE3_REQUIRE(in_channels > 0 && in_channels % kCC == 0 && out_channels > 0 && out_channels % 16 == 0, SE3_ERR_UNSUPPORTED,
              "kpconv_split_weights: channels (%d, %d) must be multiples of (8, 16)", in_channels, out_channels);
  const int64_t frags = (int64_t)(in_channels / kCC) * kKS * (out_channels / 16);
  kpconv_split_weights_kernel<<<(unsigned)frags, 64, 0, (hipStream_t)stream>>>(weights, in_channels, out_channels,
                                                                              static_cast<uint4*>(fragments));
  SE3_CHECK_LAUNCH("kpconv_split_weights");
  return SE3_OK;
}

extern "C" int se3_kpconv_so3_contract(const float* F, const void* weight_fragments, int64_t num_queries, int in_channels,
                                       int out_channels, float* out, void* stream) {
  SE3_REQUIRE(F && weight_fragments && out, SE3_ERR_INVALID_ARG, "kpconv_so3_contract: null pointer");
  SE3_REQUIRE(in_channels > 0 && in_channels % kCC == 0 && out_channels >= 16 && out_channels % 16 == 0, SE3_ERR_UNSUPPORTED,
              "kpconv_so3_contract: channels (%d, %d) must be multiples of (8, 16)", in_channels, out_channels);
  if (num_queries == 0) return SE3_OK;
  const int NT = out_channels / 16;
  const int64_t tiles = se3_cdiv(num_queries, kTP), P16 = tiles * kTP;
  // column tiles per wave / per workgroup: a workgroup covers up to 16 column tiles (4 waves x 4); wide layers with few row tiles are
  // split over the columns as well so that the grid fills the chip (the G fragments are then built once per column split)
  const uint4* Wf = static_cast<const uint4*>(weight_fragments);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds_small = (size_t)kTileFloats * 4 + (size_t)kKS * kA * 4 * sizeof(SlotEntry);
  if (NT <= 4) {
    // narrow layers: one wave per output anchor, all column tiles per wave
    const dim3 grid((unsigned)tiles, 1u);
    // anchors per wave: two halve the weight-fragment reads per MFMA (they bound the 64-column layers: 1.41 vs 1.46 ms per call at the
    // bench shape) but leave 9 waves per CU, too few to hide the build latency of the cheaper 16/32-column layers (0.92 vs 0.86 ms)
    static const char* rws = getenv("SE3_KPCONV_RW");
    const int rw = rws ? atoi(rws) : (NT == 4 ? 2 : 1);
    static const char* rdbgs = getenv("SE3_KPCONV_RDBG");
    const int rdbg = rdbgs ? atoi(rdbgs) : 0;
#define SE3_ABL(D_) if (rdbg == D_ && NT == 4) { kpconv_contract_rows_kernel<4, 2, D_><<<grid, 192, lds_small, st>>>(F, Wf, num_queries, P16, in_channels, out_channels, out); SE3_CHECK_LAUNCH("kpconv_so3_contract"); return SE3_OK; }
    SE3_ABL(1) SE3_ABL(2) SE3_ABL(4) SE3_ABL(8) SE3_ABL(3) SE3_ABL(7) SE3_ABL(15) SE3_ABL(11) SE3_ABL(6)
#undef SE3_ABL
#define SE3_ROWS(NT_)                                                                                                                \
  if (rw == 2) kpconv_contract_rows_kernel<NT_, 2><<<grid, 192, lds_small, st>>>(F, Wf, num_queries, P16, in_channels, out_channels, out); \
  else kpconv_contract_rows_kernel<NT_, 1><<<grid, 384, lds_small, st>>>(F, Wf, num_queries, P16, in_channels, out_channels, out);
    if (NT == 4) { SE3_ROWS(4) } else if (NT == 3) { SE3_ROWS(3) } else if (NT == 2) { SE3_ROWS(2) } else { SE3_ROWS(1) }
#undef SE3_ROWS
  } else {
    // wider layers: the same kernel, one workgroup per (point tile, block of 64 columns)
    SE3_REQUIRE(NT % 4 == 0, SE3_ERR_UNSUPPORTED, "kpconv_so3_contract: %d output channels (need <= 64 or a multiple of 64)", out_channels);
    // 128 columns per workgroup where the layer allows it: a built fragment then feeds 48 MFMAs instead of 24 -- 1.3 instead of 2.6 other
    // vector instructions per MFMA, which is what fits into the MFMAs' shadow (1.68 -> 1.52 and 1.50 -> 1.31 ms on the 128 / 256-column layers)
    static const char* w8 = getenv("SE3_KPCONV_NT8");
    if (NT % 8 == 0 && (w8 ? atoi(w8) != 0 : true)) {
      const dim3 grid8((unsigned)tiles, (unsigned)(NT / 8));
      kpconv_contract_rows_kernel<8, 2><<<grid8, 192, lds_small, st>>>(F, Wf, num_queries, P16, in_channels, out_channels, out);
    } else {
      const dim3 grid((unsigned)tiles, (unsigned)(NT / 4));
      kpconv_contract_rows_kernel<4, 2><<<grid, 192, lds_small, st>>>(F, Wf, num_queries, P16, in_channels, out_channels, out);
    }
  }
  SE3_CHECK_LAUNCH("kpconv_so3_contract");
  return SE3_OK;
}
