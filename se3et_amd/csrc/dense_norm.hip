// B2: the dense layer of a unary block fused with the GroupNorm on either side of it (inference path of UnaryBlockEPN,
// geotransformer/modules/e2pn/blocks_epn.py:639-665 mlp + 684-701 GroupNormEPN, as composed by ResnetBottleneckBlockEPN, :798-852).
//
//   y = T(x) W^T            T: the pending GroupNorm(s) + LeakyReLU of the producer, applied to x on its way into LDS (0, 1 or 2 stages of
//                           v -> lrelu(v scale[seg][k] + shift[seg][k])): the normalised activation is never written to HBM
//   statistics of y         Welford partials (count, mean, M2) per (row chunk, output channel) from the accumulators, merged over the
//                           chunk's row tiles in registers: the GroupNorm that follows needs no pass over y (rowops.hip: gn_partial4_kernel);
//                           the layer's bias is not added to y: it shifts the means (gn_finalize_kernel) and nothing else
//
// Round 4: the EXPANDING layers of a bottleneck block (unary2: mid -> 4 mid channels, skip_conv: in -> 2 in) no longer write their raw
// output at all.  Writing y (rows x N), reading it back in the apply pass and writing the normalised result moved 3 N floats per row for a
// GEMM whose input is N / 4 (N / 2) wide; recomputing the GEMM is cheaper than one of those passes.  Two more modes of the same kernel:
//   MODE 1 (statistics only)   the GEMM with the Welford epilogue and NO store: -> the affine table of GroupNorm(y + bias)
//   MODE 2 / 3 (final)         the GEMM again, epilogue out = lrelu(y scale + shift + R): R = a residual tensor (MODE 2: identity shortcut)
//                              or a SECOND GEMM over the shortcut's input with the shortcut norm's own table (MODE 3: skip_conv), the two
//                              products sharing one accumulator set (the first is rescaled by scale_1 / scale_2 per column in between)
// so the block's output is written once and nothing else of its width ever exists (blocks_epn.py:798-852; se3et_amd/modules/e2pn).
//
// Arithmetic as csrc/linear_f16.hip: f16 hi + lo pieces of both operands, three products on v_mfma_f32_32x32x16_f16, f32 accumulation
// (error 2^-22 per term).  One workgroup (4 waves) owns one row chunk of one segment (the chunks of the GroupNorm statistics: a chunk
// never straddles two pairs of the stacked batch) and walks its row tiles with the loads two K-steps ahead of the multiplies, across tile
// boundaries: these layers are short (K 32 .. 256 for most of the rows), their time is HBM time, and a tile-at-a-time kernel leaves the
// memory system idle during every prologue and epilogue (linear_f16_kernel: 2.4 TB/s at K = 64).
#include "common.h"
#include "group_norm.h"

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using i32x4 = __attribute__((ext_vector_type(4))) int;

// rows whose values left the headroom of their f16-split scale (or were NaN / Inf) since the last reset: se3_debug_dense_saturated_rows
__device__ unsigned long long g_dense_saturated_rows = 0;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {                      // the value of another lane of the same row of 16 (DPP control CTRL)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

constexpr int kHeaderB = 256;                 // weight pieces: se3_linear_split_weights_f16 (csrc/linear_f16.hip)
constexpr int kRowB = 80;                     // bytes per (piece, row) of the A image of one K-step: 32 f16 + 16 B pad

struct DenseArgs {
  const float* x;            // (rows, K)
  const float* in_affine[2]; // per stage: [segment][2][K] (scale, shift), or null
  float in_slope[2];
  int K, N, NCT;
  const u32x4* Wf;
  const float* hdr;
  float* out;                // (rows, N): x W^T, no bias
  float* part;               // [chunk][groups][3]: (count, mean, M2) of (out + bias) over the chunk's rows and the group's channels
  int* counters;             // [segment][column block]: chunks of the segment that have delivered their partials (zero between launches)
  const float* xb;           // (N) bias of the layer (shifts the means only), or null
  const float* gw;           // (N) GroupNorm weight, bias
  const float* gb;
  float* affine;             // out: [segment][2][N]
  int groups;
  float eps;
  SegTable T;
  // final modes (2, 3): out = lrelu(y scale + shift + R, final_slope)
  const float* affine1;      // [segment][2][N]: the table of this layer's own GroupNorm (from a MODE 1 call)
  const float* residual;     // MODE 2: (rows, N) tensor R
  const float* x2;           // MODE 3: (rows, K2) input of the shortcut layer, R = GroupNorm_2(x2 W2^T + b2) = x2 W2^T scale_2 + shift_2
  int K2;
  const u32x4* Wf2;
  const float* hdr2;
  const float* affine2;      // [segment][2][N]
  float final_slope;
  // plain modes (4, 5): out = act(T(x) W^T + bias); x rows x_rs floats apart, out rows out_rs floats apart (0: K / N)
  int x_rs, out_rs;
  int t_rows;                // MODE 5: rows per output block (anchor): out is (rows / t_rows, N, t_ld) -- the product transposed per block
  int t_ld;
  // K-segmented input (plain modes): every `seg_steps` K-steps of 32 channels come from the next of several (rows, x_rs) arrays `seg_stride`
  // floats apart -- the anchor-concatenated rows of RotCompressOutput (output_layer.py:38-47: 'b a n c -> b n (a c)') read in place from
  // (A, rows, C); seg_steps = 0: one array
  int seg_steps;
  long long seg_stride;
};


// WM x WN waves, each RT x CT MFMA tiles of 32 x 32: rows per tile TR = 32 RT WM, columns per workgroup BN = 32 CT WN.
// MODE 0: store y + statistics; 1: statistics only; 2: final with a residual tensor; 3: final with a second (shortcut) GEMM;
// 4: plain dense layer, out = act(x W^T + bias) (any K % 32 == 0, any N, row strides: the transformer's linears, se3_linear_stream);
// 5: the same with the product stored TRANSPOSED per block of t_rows rows (the value projection in the attention kernels' operand layout).
// The K-steps of a row tile are those of source 1 (K / 32) followed, in MODE 3, by those of source 2 (K2 / 32); the walk over (tile, step)
// is three uniform counters (request: three steps ahead, stage: one ahead, multiply).
struct StepPos {
  int tile, kk;
};
// (MODE 3 keeps the constants of two layers and two operand streams alive: 2 workgroups per compute unit, 256 registers, instead of 3 with
// spills -- a scratch reload inside the tile loop drains every prefetched load, DESIGN.md section 4 "RPE" lesson 2.)
template <int WM, int WN, int RT, int CT, int MODE>
__global__ __launch_bounds__(256, MODE == 3 ? 2 : 3) void dense_norm_kernel(const DenseArgs a) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TR = 32 * RT * WM, BN = 32 * CT * WN, U = TR / 32;      // U: float4 units per thread and K-step
  constexpr int kPieceB = TR * kRowB, kBufB = 2 * kPieceB;
  constexpr bool kStats = MODE <= 1, kDual = MODE == 3, kPlain = MODE >= 4;
  __shared__ __align__(16) unsigned char lds[2 * kBufB];
  // Row scales of the f16 split (the split used to be exact only for 2^-3 <= |x| < 65504, with nothing guarding the
  // window).  Every row of the A operand takes its scale from the largest magnitude of its FIRST 32 values (the first K-step of the row
  // tile): inside [2^-4, 2^7) the row is split as it is (exponent 141: error <= 2^-21 of that magnitude, 2^8 of headroom to the top of
  // f16); outside, the row is multiplied by the power of two that puts that magnitude into [2^6, 2^7) (exact), and the epilogue takes the
  // scale out of the accumulators with v_ldexp.  A row whose first 32 values are all zero stays unscaled.  The scale of a row is fixed
  // for the tile: a later value that exceeds the row's headroom (more than 2^8 times its first 32 values -- or NaN / Inf) is clamped to
  // the f16 range and COUNTED (se3_debug_dense_saturated_rows): loud instead of Inf; a NaN / Inf input value makes its row's outputs NaN
  // (both pieces NaN), as the reference's matmul would.  A row's scale is a function of that row's values
  // only: rows -- the padding rows of a packed batch included -- do not influence each other.
  __shared__ __align__(16) int row_exp[2][TR];                           // per image: exponent - 141 of a scaled row, valid for the tile ending in step ...
  __shared__ __align__(16) int row_exp_step[2][TR];                      // ... row_exp_step (any other tile: 0)
  __shared__ int row_run[TR];                                            // exponent of the rows of the tile being staged (only while a wave is off the plain path)
  __shared__ int row_scl[2];                                             // per image: the step number if some row of the tile ending in it is scaled
  extern __shared__ __align__(16) float aff[];                           // [stage][2][K] of this workgroup's segment
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int i32 = lane & 31, h = lane >> 5;
  const int K = a.K, N = a.N;
  const int K2 = kDual ? a.K2 : 32;
  const int XS = kPlain ? a.x_rs : K, OS = kPlain ? a.out_rs : N;        // row strides of x and out in floats
  long long r0, r1;
  chunk_rows(a.T, blockIdx.x, r0, r1);
  if (r1 < r0) r1 = r0;                                                  // (a chunk emptied by the row quantum: no tiles, partials that count nothing)
  const int seg = seg_of_chunk(a.T, blockIdx.x);
  int cb0 = a.T.chunk_begin[0], cb1 = a.T.chunk_begin[1];
#pragma unroll
  for (int i = 1; i < kGNMaxSegments; i++)
    if (seg == i) {
      cb0 = a.T.chunk_begin[i];
      cb1 = a.T.chunk_begin[i + 1];
    }
  cb0 = __builtin_amdgcn_readfirstlane(cb0);
  cb1 = __builtin_amdgcn_readfirstlane(cb1);
  const int stages = (a.in_affine[0] != nullptr) + (a.in_affine[1] != nullptr);
  for (int st = 0; st < stages; st++)
    for (int i = tid; i < 2 * K; i += 256) aff[st * 2 * K + i] = a.in_affine[st][(size_t)seg * 2 * K + i];
  if (tid < 2) row_scl[tid] = -1;
  for (int i = tid; i < 2 * TR; i += 256) row_exp_step[0][i] = -1;
  const int nk1 = K >> 5, nk2 = kDual ? (K2 >> 5) : 0, nkt = nk1 + nk2;
  const int ntiles = (int)((r1 - r0 + TR - 1) / TR);
  const int total = ntiles * nkt;
  auto advance = [&](StepPos& p) {
    p.kk++;
    if (p.kk == nkt) {
      p.kk = 0;
      p.tile++;
    }
  };
  // A staging: unit j of a thread = 4 consecutive floats of row (tid >> 3) + 32 j, floats 4 (tid & 7) .. + 3 of the K-step
  // Buffer addressing (uniform base and step offset, 32-bit lane offsets): rows past the end of the chunk read as zeros.
  const int q = tid & 7, urow = tid >> 3;
  const long long rbase = r1 > r0 ? r0 : 0;
  const int nrows = (int)(r1 - r0);
  const int xoff1 = (urow * XS + 4 * q) * 4, xoff2 = (urow * K2 + 4 * q) * 4;
  bool plain_rows = true;                                                // (uniform per wave) every row this wave stages is at exponent 141
  auto request = [&](const StepPos& p, f32x4 (&v)[U]) {
    const bool two = kDual && p.kk >= nk1;                               // uniform
    const int Ks = two ? K2 : XS, kloc = two ? p.kk - nk1 : p.kk;
    const float* base = two ? a.x2 + rbase * K2 : a.x + rbase * XS;
    const long long xbytes = (kPlain && a.seg_steps > 0) ? ((long long)(K / (32 * a.seg_steps) - 1) * a.seg_stride + (long long)nrows * Ks) * 4
                                                         : (long long)nrows * Ks * 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)xbytes, 0x00020000);
    int soff = (p.tile * TR * Ks + kloc * 32) * 4;
    const int xoff = two ? xoff2 : xoff1, jstep = 32 * Ks * 4;
    if constexpr (kPlain) {
      if (a.seg_steps > 0) {                                             // segment s = kloc / seg_steps: channels (kloc % seg_steps) * 32 of array s
        const int sg = kloc / a.seg_steps;
        soff = (int)(((long long)p.tile * TR * Ks + (kloc - sg * a.seg_steps) * 32 + sg * a.seg_stride) * 4);
      }
    }
#pragma unroll
    for (int j = 0; j < U; j++) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff + j * jstep, soff, 0));
  };
  auto stage = [&](int buf, const StepPos& p, const f32x4 (&v)[U]) {    // T(v) -> f16 hi / lo -> LDS image `buf`
    const bool two = kDual && p.kk >= nk1;
    const int st_here = two ? 0 : stages;                                // the shortcut's input is a concrete tensor
    const int k = (two ? 0 : p.kk) * 32 + 4 * q;
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sh0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh1 = sh0;
    if (st_here > 0) {
      sc0 = *reinterpret_cast<const f32x4*>(aff + k);
      sh0 = *reinterpret_cast<const f32x4*>(aff + K + k);
    }
    if (st_here > 1) {
      sc1 = *reinterpret_cast<const f32x4*>(aff + 2 * K + k);
      sh1 = *reinterpret_cast<const f32x4*>(aff + 3 * K + k);
    }
    // (plain modes with x_rs < K: a row is read past its end into the next one against zero weight columns -- those values are dropped
    // here, not multiplied: an Inf / NaN / > 65504 in the neighbouring row would turn 0 * x into NaN.  Uniform condition.)
    const bool tail_mask = kPlain && XS < K && a.seg_steps == 0 && k + 4 > XS;
    const bool first_step = p.kk == 0, last_step = p.kk == nkt - 1;      // (uniform) of the row tile
    auto transformed = [&](const f32x4& raw) {                          // T(v): tail mask, the pending norm stages
      f32x4 t = raw;
      if constexpr (kPlain) {
        if (tail_mask) {
#pragma unroll
          for (int e = 0; e < 4; e++) t[e] = k + e < XS ? t[e] : 0.f;
        }
      }
      if (st_here > 0) {
        t = t * sc0 + sh0;
#pragma unroll
        for (int e = 0; e < 4; e++) t[e] = t[e] > 0.f ? t[e] : t[e] * a.in_slope[0];
      }
      if (st_here > 1) {
        t = t * sc1 + sh1;
#pragma unroll
        for (int e = 0; e < 4; e++) t[e] = t[e] > 0.f ? t[e] : t[e] * a.in_slope[1];
      }
      return t;
    };
    // One decision per call, wave-uniform, from four vector instructions per unit (two maxima, two compares; the rest is scalar): every row this
    // wave stages is at exponent 141, nothing of this K-step reaches 2^15 (first step: 2^7) and -- in a tile's first step -- every row holds
    // something at 2^-4 or above.  That is the straight-line common path (the unscaled split); anything else takes the exact path.
    unsigned long long odd = 0;                                          // lanes that see a value at / above 2^15 (NaN included) [or, first step, a row below 2^-4]
    f32x4 tv[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const f32x4 t = transformed(v[j]);
      tv[j] = t;
      const float m = fmaxf(fmaxf(__builtin_fabsf(t[0]), __builtin_fabsf(t[1])), fmaxf(__builtin_fabsf(t[2]), __builtin_fabsf(t[3])));
      odd |= __builtin_amdgcn_ballot_w64(!(m < (first_step ? 128.f : 32768.f)));
      // a row = 8 adjacent lanes = one byte of the ballot: the bytes are OR-folded on the scalar unit, a zero byte is a row below 2^-4
      unsigned long long ge = __builtin_amdgcn_ballot_w64(m >= 0.0625f);
      ge |= ge >> 4;
      ge |= ge >> 2;
      ge |= ge >> 1;
      odd |= first_step ? (~ge & 0x0101010101010101ull) : 0ull;
    }
    auto split_store = [&](int j, const f32x4& t) {
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        hi[e] = (_Float16)t[e];
        lo[e] = (_Float16)(t[e] - (float)hi[e]);
      }
      unsigned char* dst = lds + buf * kBufB + (urow + 32 * j) * kRowB + q * 8;
      *reinterpret_cast<f16x4*>(dst) = hi;
      *reinterpret_cast<f16x4*>(dst + kPieceB) = lo;
    };
    if (__builtin_expect(odd == 0 && (plain_rows || first_step), 1)) {
#pragma unroll
      for (int j = 0; j < U; j++) split_store(j, tv[j]);
      plain_rows = true;
      return;
    }
    // ---- the exact path: per row the largest magnitude of this K-step -> (first step) the row's scale, (later) the saturation check.  NOT
    // unrolled over the units (the unit is picked with selects): the code of this cold path sits inside the hot loop, and its size alone
    // costs the loop time (measured: the same branch, never taken, with the unrolled body: +5 % on the unary shapes).
    bool any_scaled = false;
    int saturated = 0;
#pragma unroll 1
    for (int j = 0; j < U; j++) {
      f32x4 t = tv[0];
#pragma unroll
      for (int jj = 1; jj < U; jj++) t = j == jj ? tv[jj] : t;
      float m = fmaxf(fmaxf(__builtin_fabsf(t[0]), __builtin_fabsf(t[1])), fmaxf(__builtin_fabsf(t[2]), __builtin_fabsf(t[3])));
      const bool nan = (t[0] != t[0]) | (t[1] != t[1]) | (t[2] != t[2]) | (t[3] != t[3]);
      m = nan ? __builtin_bit_cast(float, 0x7fc00000) : m;              // (fmaxf drops NaN)
      int mb = __builtin_bit_cast(int, m);                               // non-negative floats order like their bit patterns, NaN last
      mb = max(mb, __builtin_amdgcn_update_dpp(0, mb, 0xB1, 0xf, 0xf, true));     // the row's 32 values sit in 8 adjacent lanes: quad_perm [1, 0, 3, 2],
      mb = max(mb, __builtin_amdgcn_update_dpp(0, mb, 0x4E, 0xf, 0xf, true));     // quad_perm [2, 3, 0, 1],
      mb = max(mb, __builtin_amdgcn_update_dpp(0, mb, 0x141, 0xf, 0xf, true));    // row_half_mirror
      const int e_here = (mb >> 23) & 0xff;                              // biased exponent of the row's largest magnitude in this step
      const int row = urow + 32 * j;
      int e_row;
      if (first_step)        // plain inside [2^-4, 2^7) (exponents 123 .. 133) and for an all-zero start; else the magnitude goes to [2^6, 2^7)
        e_row = ((e_here >= 123 && e_here <= 133) || e_here == 0) ? 141 : (e_here >= 255 ? 141 : min(e_here + 8, 254));
      else
        e_row = plain_rows ? 141 : row_run[row];
      if (first_step && q == 0) row_run[row] = e_row;
      const bool sat = e_here > e_row;                                   // beyond the row's headroom (NaN / Inf: 255)
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float te = __builtin_ldexpf(t[e], 141 - e_row);
        te = fminf(fmaxf(te, -65000.f), 65000.f);                        // (finite overflow of the row's headroom: clamped, and counted below)
        te = (__float_as_uint(t[e]) & 0x7f800000u) == 0x7f800000u ? __builtin_bit_cast(float, 0x7fc00000) : te;   // NaN / Inf: the row's outputs are NaN
        hi[e] = (_Float16)te;
        lo[e] = (_Float16)(te - (float)hi[e]);
      }
      unsigned char* dst = lds + buf * kBufB + row * kRowB + q * 8;
      *reinterpret_cast<f16x4*>(dst) = hi;
      *reinterpret_cast<f16x4*>(dst + kPieceB) = lo;
      if (last_step && e_row != 141 && q == 0) {
        row_exp[buf][row] = e_row - 141;
        row_exp_step[buf][row] = p.tile * nkt + p.kk;
      }
      saturated += (sat && q == 0) ? 1 : 0;
      any_scaled |= e_row != 141;
    }
    const bool w_scaled = __builtin_amdgcn_ballot_w64(any_scaled) != 0;
    plain_rows = !w_scaled;
    if (last_step && w_scaled && lane == 0) row_scl[buf] = p.tile * nkt + p.kk;      // (every writer stores the same number)
    if (saturated) atomicAdd(&g_dense_saturated_rows, (unsigned long long)saturated);
  };
  const int a_read = (wm * (RT * 32) + i32) * kRowB + h * 16;           // + rt * 32 * kRowB + ks * 32 + piece * kPieceB
  const int ct0 = blockIdx.y * (BN / 32) + wn * CT;
  const u32x4* wbase1 = a.Wf + (int64_t)ct0 * 2 * 64 + lane;
  const u32x4* wbase2 = kDual ? a.Wf2 + (int64_t)ct0 * 2 * 64 + lane : wbase1;
  int coff[CT];                                                          // fragment offset of this wave's column tile c (plain modes: N need
#pragma unroll                                                           // not fill the last column block -- tiles past the end re-read the last one)
  for (int c = 0; c < CT; c++) coff[c] = kPlain ? ((ct0 + c < a.NCT ? ct0 + c : a.NCT - 1) - ct0) * 128 : c * 128;
  const int64_t wstep = (int64_t)a.NCT * 2 * 64;                        // u32x4 per K16-step (the same for both sources: same N)
  const float inv_scale = a.hdr[0];
  // final modes: per-column constants of this lane's CT columns
  float fs[CT], bs[CT], rescale[CT];
#pragma unroll
  for (int c = 0; c < CT; c++) fs[c] = bs[c] = rescale[c] = 1.f;
  if constexpr (kPlain) {
#pragma unroll
    for (int c = 0; c < CT; c++) {
      const int col = (ct0 + c) * 32 + i32;
      fs[c] = inv_scale;
      bs[c] = (a.xb != nullptr && col < N) ? a.xb[col] : 0.f;
    }
  } else if constexpr (MODE >= 2) {
#pragma unroll
    for (int c = 0; c < CT; c++) {
      const int col = (ct0 + c) * 32 + i32;
      const float a1 = a.affine1[(size_t)seg * 2 * N + col], b1 = a.affine1[(size_t)seg * 2 * N + N + col];
      if constexpr (kDual) {
        // acc = S1 G1 after source 1 (S: the power-of-two scale of the weight pieces); rescaled to S2 G1 a1 / a2 it takes S2 G2 on top,
        // and out = acc a2 / S2 + (b1 + b2).  (a2 = 0 has no such form: the host keeps layers with a zero norm weight off this path.)
        const float a2 = a.affine2[(size_t)seg * 2 * N + col], b2 = a.affine2[(size_t)seg * 2 * N + N + col];
        const float inv2 = a.hdr2[0];
        rescale[c] = (a1 / a2) * (inv_scale / inv2);
        fs[c] = a2 * inv2;
        bs[c] = b1 + b2;
      } else {
        fs[c] = a1 * inv_scale;
        bs[c] = b1;
      }
    }
  }
  f32x16 acc[RT][CT];
#pragma unroll
  for (int r = 0; r < RT; r++)
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[r][c][v] = 0.f;
  WF run[CT];
#pragma unroll
  for (int c = 0; c < CT; c++) run[c] = {0.f, 0.f, 0.f};
  // Weight fragments of the next WD K-steps (two K16 sub-steps each), a ring of register sets: set s % WD holds K-step s and is refilled
  // with K-step s + WD behind the MFMAs that read it.  One set (the fragments requested one K-step = 100-200 ns of MFMAs ahead of an L2
  // round trip of ~700) is enough where three workgroups per compute unit walk many row tiles; the transformer's layers are ONE tile per
  // workgroup on fewer workgroups than compute units -- a chain of K / 32 round trips -- and their small tiles leave the registers for more.
  constexpr int WD = (kPlain && RT == 1 && WM == 2) ? (CT == 1 ? 4 : 2) : 1;
  u32x4 bq[WD][2][CT][2];
  auto weights_of = [&](int kstep) -> const u32x4* {                      // fragments of K-step `kstep` of the tile's sequence (uniform)
    const bool two = kDual && kstep >= nk1;
    return (two ? wbase2 : wbase1) + (int64_t)(2 * (two ? kstep - nk1 : kstep)) * wstep;
  };
#pragma unroll
  for (int d = 0; d < WD; d++) {
    const u32x4* w0 = weights_of(d % nkt);
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int c = 0; c < CT; c++) {
        bq[d][j][c][0] = w0[j * wstep + coff[c]];
        bq[d][j][c][1] = w0[j * wstep + coff[c] + 64];
      }
  }

  auto multiply = [&](int buf, const StepPos& p, u32x4 (&bq)[2][CT][2]) {
    if constexpr (kDual) {
      if (p.kk == nk1) {                                                 // between the two products: S1 G1 -> S2 G1 a1 / a2
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
          for (int c = 0; c < CT; c++) acc[r][c] *= rescale[c];
      }
    }
    const unsigned char* img = lds + buf * kBufB + a_read;
    // weight fragments of K-step + WD go where this step's were (the sequence wraps at a tile end)
    int kn = p.kk + WD;
    while (kn >= nkt) kn -= nkt;
    const u32x4* wnext = weights_of(kn);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      f16x8 av[RT][2];
#pragma unroll
      for (int r = 0; r < RT; r++) {
        av[r][0] = *reinterpret_cast<const f16x8*>(img + r * 32 * kRowB + ks * 32);
        av[r][1] = *reinterpret_cast<const f16x8*>(img + r * 32 * kRowB + ks * 32 + kPieceB);
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const f16x8 b0 = __builtin_bit_cast(f16x8, bq[ks][c][0]);
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[r][1], b0, acc[r][c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const f16x8 b1 = __builtin_bit_cast(f16x8, bq[ks][c][1]);
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[r][0], b1, acc[r][c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const f16x8 b0 = __builtin_bit_cast(f16x8, bq[ks][c][0]);
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[r][0], b0, acc[r][c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        bq[ks][c][0] = wnext[ks * wstep + coff[c]];
        bq[ks][c][1] = wnext[ks * wstep + coff[c] + 64];
      }
    }
    if (p.kk != nkt - 1) return;
    // ---- end of a row tile
    const long long trow0 = r0 + (long long)p.tile * TR + wm * (RT * 32);          // uniform
    const int rows_here = (int)(r1 - trow0 < RT * 32 ? (r1 - trow0 > 0 ? r1 - trow0 : 0) : RT * 32);
    // the row scales out again: acc 2^(E - 141), exact (v_ldexp: gradual underflow instead of a flushed factor); tiles without a scaled
    // row -- all of them, for activations of ordinary magnitude -- skip this on one flag
    if (__builtin_expect(row_scl[buf] == p.tile * nkt + p.kk, 0)) {
      // (round 6: the exponents of a lane's four consecutive rows as ONE 16-byte read each of the step tags and of the exponents, selected in
      //  registers.  The round-4 form -- a conditional 4-byte read per accumulator row between scheduling barriers, spilled through scratch --
      //  scaled accumulator row 1 of a tile by a stale word whenever ANOTHER row of the tile carried a scale: one row in ~10^5 of the
      //  backbone came out as Inf, which the f16 split of the next layer clamped to 65000 unnoticed until non-finite values were allowed
      //  to propagate; tests/test_gpu_ops.py::test_a_scaled_row_leaves_the_other_rows_of_its_tile_alone.)
      const int step_id = p.tile * nkt + p.kk;
#pragma unroll
      for (int r = 0; r < RT; r++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int base = (wm * RT + r) * 32 + 8 * g + 4 * h;              // rows base .. base + 3 = accumulator registers 4 g .. 4 g + 3
          const i32x4 tag = *reinterpret_cast<const i32x4*>(&row_exp_step[buf][base]);
          const i32x4 ex = *reinterpret_cast<const i32x4*>(&row_exp[buf][base]);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int ev = tag[j] == step_id ? ex[j] : 0;
#pragma unroll
            for (int c = 0; c < CT; c++) acc[r][c][4 * g + j] = __builtin_ldexpf(acc[r][c][4 * g + j], ev);
          }
        }
    }
    if constexpr (MODE == 4) {
      // out = act(acc / S + bias): rows past the chunk and columns past N get an out-of-range offset (dropped by the buffer bounds check:
      // the store instruction itself stays unconditional, so the number of outstanding memory operations does not depend on the path)
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out + trow0 * OS, 0, rows_here > 0 ? ((rows_here - 1) * OS + N) * 4 : 0, 0x00020000);
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const int col = (ct0 + c) * 32 + i32;
        const int lane_off = col < N ? (4 * h * OS + col) * 4 : 0x7ffffff0;
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
          for (int v = 0; v < 16; v++) {
            const int rr = r * 32 + (v & 3) + 8 * (v >> 2);
            float val = acc[r][c][v] * fs[c] + bs[c];
            val = fmaxf(val, val * a.final_slope);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ors, lane_off, rr * OS * 4, 0);
            acc[r][c][v] = 0.f;
          }
      }
      return;
    }
    if constexpr (MODE == 5) {
      // transposed per block of t_rows rows: out[(row / t_rows) N + col][row % t_rows]; a lane's four consecutive rows are one 16-byte store
      // (t_rows % 4 == 0, tiles start at multiples of 32 rows: a group of four never straddles a block or the end of the rows)
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const int col = (ct0 + c) * 32 + i32;
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const long long row = trow0 + r * 32 + 8 * g + 4 * h;
            const int blk = (int)(row / a.t_rows), rloc = (int)(row - (long long)blk * a.t_rows);
            f32x4 v4;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              float val = acc[r][c][4 * g + j] * fs[c] + bs[c];
              v4[j] = fmaxf(val, val * a.final_slope);
              acc[r][c][4 * g + j] = 0.f;
            }
            const bool ok = col < N && row < r1;
            const int off = ok ? (int)((((long long)blk * N + col) * a.t_ld + rloc) * 4) : 0x7ffffff0;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), ors, off, 0, 0);
          }
      }
      return;
    }
    if constexpr (MODE >= 2) {
      // out = lrelu(y scale + shift + R): the rows of this wave's block as buffers of rows_here rows (rows past the chunk read as zeros
      // and are dropped on the store)
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out + trow0 * N, 0, rows_here * N * 4, 0x00020000);
      const bool has_res = MODE == 2 && a.residual != nullptr;            // (no residual: an empty buffer, every load returns zero)
      const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_res ? a.residual + trow0 * N : a.out), 0,
                                                                           has_res ? rows_here * N * 4 : 0, 0x00020000);
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const int lane_off = (4 * h * N + (ct0 + c) * 32 + i32) * 4;
#pragma unroll
        for (int r = 0; r < RT; r++) {
          float res[16];
          if constexpr (MODE == 2) {
#pragma unroll
            for (int v = 0; v < 16; v++)
              res[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, lane_off, (r * 32 + (v & 3) + 8 * (v >> 2)) * N * 4, 0));
          }
#pragma unroll
          for (int v = 0; v < 16; v++) {
            const int rr = r * 32 + (v & 3) + 8 * (v >> 2);
            float val = acc[r][c][v] * fs[c] + bs[c];
            if constexpr (MODE == 2) val += res[v];
            val = fmaxf(val, val * a.final_slope);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ors, lane_off, rr * N * 4, 0);
            acc[r][c][v] = 0.f;
          }
        }
      }
      return;
    }
    // MODE 0 / 1: [store y,] fold the tile into the running statistics of this wave's columns, clear the accumulators
    const int left = rows_here - 4 * h;                                            // rows of this lane's column slice that exist
    // the rows of this wave's block as a buffer of rows_here rows: stores to rows past the chunk fall outside and are dropped
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(MODE == 0 ? a.out + trow0 * N : nullptr, 0, MODE == 0 ? rows_here * N * 4 : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < CT; c++) {
      const int lane_off = (4 * h * N + (ct0 + c) * 32 + i32) * 4;
      float cnt = 0.f, sum = 0.f;
#pragma unroll
      for (int r = 0; r < RT; r++)
#pragma unroll
        for (int v = 0; v < 16; v++) {
          const int rr = r * 32 + (v & 3) + 8 * (v >> 2);
          const float val = acc[r][c][v] * inv_scale;
          acc[r][c][v] = val;
          if constexpr (MODE == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ors, lane_off, rr * N * 4, 0);
          if (rr < left) {
            cnt += 1.f;
            sum += val;
          }
        }
      cnt += __shfl_xor(cnt, 32);
      sum += __shfl_xor(sum, 32);
      const float mean = sum / fmaxf(cnt, 1.f);
      float m2 = 0.f;
#pragma unroll
      for (int r = 0; r < RT; r++)
#pragma unroll
        for (int v = 0; v < 16; v++) {
          const int rr = r * 32 + (v & 3) + 8 * (v >> 2);
          const float d = acc[r][c][v] - mean;
          if (rr < left) m2 += d * d;
          acc[r][c][v] = 0.f;
        }
      m2 += __shfl_xor(m2, 32);
      run[c] = wf_merge(run[c], WF{cnt, mean, m2});
    }
  };

  f32x4 ra[U], rb[U];                                                   // the operands of steps s + 1 and s + 2
  StepPos pr{0, 0}, ps{0, 0}, pm{0, 0};                                  // request / stage / multiply positions
  request(pr, ra);
  advance(pr);
  request(pr, rb);
  advance(pr);
  __syncthreads();                                                       // the affine table
  stage(0, ps, ra);
  advance(ps);
#pragma unroll
  for (int j = 0; j < U; j++) ra[j] = rb[j];
  request(pr, rb);
  advance(pr);
  __syncthreads();
  if constexpr (WD == 1) {
#pragma unroll 1
    for (int s = 0; s < total; s++) {
      multiply(s & 1, pm, bq[0]);
      advance(pm);
      stage((s + 1) & 1, ps, ra);
      advance(ps);
#pragma unroll
      for (int j = 0; j < U; j++) ra[j] = rb[j];
      request(pr, rb);
      advance(pr);
      __syncthreads();
    }
  } else {                                                               // WD (even) steps per trip: static ring and image indices
#pragma unroll 1
    for (int s = 0; s < total; s += WD) {
#pragma unroll
      for (int j = 0; j < WD; j++) {
        if (s + j >= total) break;                                       // (uniform)
        multiply(j & 1, pm, bq[j]);
        advance(pm);
        stage((j + 1) & 1, ps, ra);
        advance(ps);
#pragma unroll
        for (int u = 0; u < U; u++) ra[u] = rb[u];
        request(pr, rb);
        advance(pr);
        __syncthreads();
      }
    }
  }
  if constexpr (!kStats) return;
  // ---- the chunk's statistics: + bias, waves that share columns merged through LDS, channels merged into their groups across lanes
  __syncthreads();
  const int cpg = N / a.groups;                                          // channels per group: a power of two <= 32 (host-checked)
  WF* sh = reinterpret_cast<WF*>(lds);                                   // [wave][CT][32]
  if (a.xb)
#pragma unroll
    for (int c = 0; c < CT; c++) run[c].mean += a.xb[(ct0 + c) * 32 + i32];
  if (h == 0)
#pragma unroll
    for (int c = 0; c < CT; c++) sh[(wave * CT + c) * 32 + i32] = run[c];
  __syncthreads();
  if (wm == 0)
#pragma unroll
    for (int c = 0; c < CT; c++) {
      WF w = run[c];
      for (int m = 1; m < WM; m++) w = wf_merge(w, sh[((m * WN + wn) * CT + c) * 32 + i32]);
      for (int d = 1; d < cpg; d <<= 1) {
        const WF o = {__shfl_xor(w.n, d), __shfl_xor(w.mean, d), __shfl_xor(w.m2, d)};
        w = wf_merge(w, o);
      }
      const int col = (ct0 + c) * 32 + i32;
      if (h == 0 && (col & (cpg - 1)) == 0) gn_store_partial(a.part + ((int64_t)blockIdx.x * a.groups + col / cpg) * 3, w);
    }
  // ---- the last chunk of a segment to arrive turns the segment's partials into the affine table
  if (!gn_last_arrival(a.counters + seg * kGNMaxColumnBlocks + blockIdx.y, cb1 - cb0)) return;
  const int gpb = BN / cpg;                                              // groups of this column block
  gn_finalize_groups(a.part, a.groups, cb0, cb1, blockIdx.y * gpb, gpb, cpg, a.xb, a.gw, a.gb, a.eps, a.affine + (size_t)seg * 2 * N, N);
}

int g_target_chunks = 768;                   // workgroups (row chunks x column blocks) aimed at: 3 per compute unit, all resident at once

}  // namespace

// Rows (of any dense launch of this process) whose values left the headroom of their f16-split scale, or held NaN / Inf, since the last
// reset: they were clamped to the f16 range instead of producing Inf.  Synchronises the device.  reset != 0: back to zero.
extern "C" unsigned long long se3_debug_dense_saturated_rows(int reset) {
  unsigned long long n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_dense_saturated_rows), sizeof(n)) != hipSuccess) return ~0ull;
  if (reset) {
    const unsigned long long zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dense_saturated_rows), &zero, sizeof(zero));
  }
  return n;
}

extern "C" void se3_dense_norm_set_target_chunks(int workgroups) { g_target_chunks = workgroups > 0 ? workgroups : 768; }

constexpr size_t kCounterB = kGNCounterB;

// [arrival counters: zero before the first call, left zero by every call][partials: (chunk, group) x 3 floats]
extern "C" size_t se3_dense_norm_workspace_bytes(int groups) {
  return kCounterB + (size_t)(kGNMaxChunks + kGNMaxSegments) * (groups > 0 ? groups : 1) * 3 * sizeof(float);
}

namespace {
// row chunks: whole row tiles, about g_target_chunks workgroups in all, never across a segment boundary
int dense_chunks(int64_t rows, int N, const int64_t* segment_row_offsets_host, int num_segments, SegTable& T, int& TR, int& BN, int& ncb,
                 bool given = false) {
  if (!given) {
    TR = N >= 256 ? 64 : 128;
    BN = N >= 256 ? 256 : N;
  }
  ncb = (N + BN - 1) / BN;
  T = SegTable{};
  T.n = num_segments;
  T.quantum = TR;
  int64_t tiles = 0;
  for (int s = 0; s < num_segments; s++) {
    const int64_t b0 = num_segments == 1 ? 0 : segment_row_offsets_host[s], b1 = num_segments == 1 ? rows : segment_row_offsets_host[s + 1];
    if (!(b1 > b0 && b0 >= 0 && b1 <= rows)) return -1;
    tiles += se3_cdiv(b1 - b0, TR);
  }
  int64_t want = g_target_chunks / ncb;
  if (want > kGNMaxChunks - num_segments) want = kGNMaxChunks - num_segments;
  if (want < 1) want = 1;
  int64_t tiles_per_chunk = se3_cdiv(tiles, want);
  for (int s = 0; s < num_segments; s++) {                                // at most 256 chunks per segment: the in-kernel finalize reads them all
    const int64_t b0 = num_segments == 1 ? 0 : segment_row_offsets_host[s], b1 = num_segments == 1 ? rows : segment_row_offsets_host[s + 1];
    const int64_t need = se3_cdiv(se3_cdiv(b1 - b0, TR), 256);
    if (need > tiles_per_chunk) tiles_per_chunk = need;
  }
  int chunks = 0;
  for (int s = 0; s < num_segments; s++) {
    const int64_t b0 = num_segments == 1 ? 0 : segment_row_offsets_host[s], b1 = num_segments == 1 ? rows : segment_row_offsets_host[s + 1];
    T.row_begin[s] = b0;
    T.row_begin[s + 1] = b1;
    T.chunk_begin[s] = chunks;
    chunks += (int)se3_cdiv(se3_cdiv(b1 - b0, TR), tiles_per_chunk);
    T.chunk_begin[s + 1] = chunks;
  }
  if (!(T.row_begin[0] == 0 && T.row_begin[num_segments] == rows && chunks <= kGNMaxChunks + kGNMaxSegments)) return -1;
  return chunks;
}

template <int MODE>
void dense_launch(const DenseArgs& a, int N, dim3 grid, size_t dyn, hipStream_t st) {
  if (N >= 256) dense_norm_kernel<1, 4, 2, 2, MODE><<<grid, 256, dyn, st>>>(a);
  else if (N == 128) dense_norm_kernel<2, 2, 2, 2, MODE><<<grid, 256, dyn, st>>>(a);
  else if (N == 64) dense_norm_kernel<4, 1, 1, 2, MODE><<<grid, 256, dyn, st>>>(a);
  else dense_norm_kernel<4, 1, 1, 1, MODE><<<grid, 256, dyn, st>>>(a);
}
}  // namespace

// out == NULL: statistics only (MODE 1) -- the GEMM runs, its result is reduced to the affine table and never stored.
extern "C" int se3_dense_norm_fwd(const float* x, int64_t rows, int in_features, const float* in_affine_a, float in_slope_a,
                                  const float* in_affine_b, float in_slope_b, const void* weight_pieces, int out_features,
                                  const float* linear_bias, const float* norm_weight, const float* norm_bias, int groups, float eps,
                                  const int64_t* segment_row_offsets_host, int num_segments, float* out, float* affine_out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(x && weight_pieces && norm_weight && norm_bias && affine_out && workspace, SE3_ERR_INVALID_ARG, "dense_norm: null pointer");
  const int K = in_features, N = out_features;
  SE3_REQUIRE(K >= 32 && K <= 1024 && (K & (K - 1)) == 0 && N >= 32 && N % 32 == 0 && (N < 256 ? (N & (N - 1)) == 0 : N % 256 == 0),
              SE3_ERR_UNSUPPORTED, "dense_norm: %d -> %d features (in: a power of two 32..1024; out: 32, 64, 128 or a multiple of 256)", K, N);
  SE3_REQUIRE(rows >= 1 && groups >= 1 && N % groups == 0, SE3_ERR_INVALID_ARG, "dense_norm: rows %lld groups %d", (long long)rows, groups);
  {
    const int cpg = N / groups;
    SE3_REQUIRE(cpg <= 32 && (cpg & (cpg - 1)) == 0 && N / (N >= 256 ? 256 : N) <= kGNMaxColumnBlocks, SE3_ERR_UNSUPPORTED,
                "dense_norm: %d channels per group (a power of two <= 32), %d output features (<= %d)", cpg, N, 256 * kGNMaxColumnBlocks);
  }
  SE3_REQUIRE(num_segments >= 1 && num_segments <= kGNMaxSegments && (num_segments == 1 || segment_row_offsets_host), SE3_ERR_UNSUPPORTED,
              "dense_norm: %d segments (1..%d)", num_segments, kGNMaxSegments);
  SE3_REQUIRE(((uintptr_t)x & 15) == 0 && (in_affine_a || !in_affine_b), SE3_ERR_INVALID_ARG, "dense_norm: x must be 16-byte aligned; stage b needs stage a");
  SE3_REQUIRE(in_slope_a >= 0.f && in_slope_a <= 1.f && in_slope_b >= 0.f && in_slope_b <= 1.f, SE3_ERR_INVALID_ARG, "dense_norm: LeakyReLU slopes must lie in [0, 1]");
  SE3_REQUIRE(workspace_bytes >= se3_dense_norm_workspace_bytes(groups), SE3_ERR_WORKSPACE, "dense_norm: workspace too small");
  SegTable T;
  int TR, BN, ncb;
  const int chunks = dense_chunks(rows, N, segment_row_offsets_host, num_segments, T, TR, BN, ncb);
  SE3_REQUIRE(chunks > 0, SE3_ERR_INVALID_ARG, "dense_norm: the segments must be non-empty, ascending and cover all %lld rows", (long long)rows);
  DenseArgs a{};
  a.x = x;
  a.in_affine[0] = in_affine_a;
  a.in_affine[1] = in_affine_b;
  a.in_slope[0] = in_slope_a;
  a.in_slope[1] = in_slope_b;
  a.K = K;
  a.N = N;
  a.NCT = (N + 63) / 64 * 2;
  a.hdr = static_cast<const float*>(weight_pieces);
  a.Wf = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces) + kHeaderB);
  a.out = out;
  a.counters = static_cast<int*>(workspace);
  a.part = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + kCounterB);
  a.xb = linear_bias;
  a.gw = norm_weight;
  a.gb = norm_bias;
  a.affine = affine_out;
  a.groups = groups;
  a.eps = eps;
  a.T = T;
  const size_t dyn = (size_t)((in_affine_a != nullptr) + (in_affine_b != nullptr)) * 2 * K * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)chunks, (unsigned)ncb);
  if (out != nullptr) dense_launch<0>(a, N, grid, dyn, st);
  else dense_launch<1>(a, N, grid, dyn, st);
  SE3_CHECK_LAUNCH("dense_norm");
  return SE3_OK;
}

// The tail of a bottleneck block (blocks_epn.py:838-852: x = unary2(x); return lrelu(x + shortcut)) with nothing of the output's width
// written but the output: out = lrelu( GroupNorm(T(x) W^T + b) + R ), the GroupNorm given by its table `affine` (a statistics-only
// se3_dense_norm_fwd call on the same operands), R = `residual` (rows, out_features), or GroupNorm_2(x2 W2^T + b2) of the shortcut layer
// given by (x2, weight_pieces2, affine2), or nothing.
extern "C" int se3_dense_residual_fwd(const float* x, int64_t rows, int in_features, const float* in_affine_a, float in_slope_a,
                                      const float* in_affine_b, float in_slope_b, const void* weight_pieces, const float* affine,
                                      const float* x2, int in_features2, const void* weight_pieces2, const float* affine2,
                                      const float* residual, int out_features, float final_slope, const int64_t* segment_row_offsets_host,
                                      int num_segments, float* out, void* stream) {
  SE3_REQUIRE(x && weight_pieces && affine && out, SE3_ERR_INVALID_ARG, "dense_residual: null pointer");
  const int K = in_features, N = out_features, K2 = in_features2;
  SE3_REQUIRE(K >= 32 && K <= 1024 && (K & (K - 1)) == 0 && N >= 32 && N % 32 == 0 && (N < 256 ? (N & (N - 1)) == 0 : N % 256 == 0),
              SE3_ERR_UNSUPPORTED, "dense_residual: %d -> %d features (in: a power of two 32..1024; out: 32, 64, 128 or a multiple of 256)", K, N);
  SE3_REQUIRE(!(x2 && residual), SE3_ERR_INVALID_ARG, "dense_residual: a residual tensor OR a shortcut layer");
  SE3_REQUIRE(!x2 || (weight_pieces2 && affine2 && K2 >= 32 && K2 <= 1024 && (K2 & (K2 - 1)) == 0 && ((uintptr_t)x2 & 15) == 0), SE3_ERR_UNSUPPORTED,
              "dense_residual: shortcut layer with %d input features (a power of two 32..1024, 16-byte aligned rows)", K2);
  SE3_REQUIRE(rows >= 1 && num_segments >= 1 && num_segments <= kGNMaxSegments && (num_segments == 1 || segment_row_offsets_host), SE3_ERR_UNSUPPORTED,
              "dense_residual: rows %lld, %d segments (1..%d)", (long long)rows, num_segments, kGNMaxSegments);
  SE3_REQUIRE(((uintptr_t)x & 15) == 0 && (in_affine_a || !in_affine_b), SE3_ERR_INVALID_ARG, "dense_residual: x must be 16-byte aligned; stage b needs stage a");
  SE3_REQUIRE(in_slope_a >= 0.f && in_slope_a <= 1.f && in_slope_b >= 0.f && in_slope_b <= 1.f && final_slope >= 0.f && final_slope <= 1.f, SE3_ERR_INVALID_ARG,
              "dense_residual: LeakyReLU slopes must lie in [0, 1]");
  SegTable T;
  int TR, BN, ncb;
  const int chunks = dense_chunks(rows, N, segment_row_offsets_host, num_segments, T, TR, BN, ncb);
  SE3_REQUIRE(chunks > 0, SE3_ERR_INVALID_ARG, "dense_residual: the segments must be non-empty, ascending and cover all %lld rows", (long long)rows);
  DenseArgs a{};
  a.x = x;
  a.in_affine[0] = in_affine_a;
  a.in_affine[1] = in_affine_b;
  a.in_slope[0] = in_slope_a;
  a.in_slope[1] = in_slope_b;
  a.K = K;
  a.N = N;
  a.NCT = (N + 63) / 64 * 2;
  a.hdr = static_cast<const float*>(weight_pieces);
  a.Wf = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces) + kHeaderB);
  a.out = out;
  a.T = T;
  a.affine1 = affine;
  a.residual = residual;
  a.final_slope = final_slope;
  const size_t dyn = (size_t)((in_affine_a != nullptr) + (in_affine_b != nullptr)) * 2 * K * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)chunks, (unsigned)ncb);
  if (x2 != nullptr) {
    a.x2 = x2;
    a.K2 = K2;
    a.hdr2 = static_cast<const float*>(weight_pieces2);
    a.Wf2 = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces2) + kHeaderB);
    a.affine2 = affine2;
    dense_launch<3>(a, N, grid, dyn, st);
  } else {
    dense_launch<2>(a, N, grid, dyn, st);      // (residual == NULL: the kernel reads an empty buffer = zeros)
  }
  SE3_CHECK_LAUNCH("dense_residual");
  return SE3_OK;
}


// ---- the same streaming kernel as a plain dense layer (the transformer's nn.Linear layers: rpe_transformer.py:56-73,134-165,
// vanilla_transformer.py:22-37, output_layer.py:7-47; geotransformer.py:213-317 in_proj / out_proj) -----------------------------------------
// out = act(x W^T + bias) for ANY row count: small products (a few thousand rows: the invariant layers, one pair per forward) take 64 x 128
// or 64 x 64 tiles so that every compute unit gets a workgroup, large ones 64 x 256; the loads run two K-steps ahead across tile boundaries.
namespace {
template <int WM, int WN, int RT, int CT>
void plain_launch(const DenseArgs& a, bool transposed, dim3 grid, hipStream_t st) {
  if (transposed) dense_norm_kernel<WM, WN, RT, CT, 5><<<grid, 256, 0, st>>>(a);
  else dense_norm_kernel<WM, WN, RT, CT, 4><<<grid, 256, 0, st>>>(a);
}
int linear_stream_one(const float* x, int64_t rows, int K, int64_t x_rs, const void* weight_pieces, const float* bias, int N, int relu, float* out,
                      int64_t out_rs, int t_rows, int64_t t_ld, void* stream, int seg_channels = 0, int64_t seg_stride = 0) {
  SE3_REQUIRE(x && weight_pieces && out, SE3_ERR_INVALID_ARG, "linear_stream: null pointer");
  SE3_REQUIRE(K > 0 && K % 32 == 0 && N > 0, SE3_ERR_UNSUPPORTED, "linear_stream: in_features %d must be a multiple of 32", K);
  // (x_rs < K is allowed: every row is then read past its end into the next one -- for weights whose columns beyond x_rs are zero; the
  // last row reads zeros beyond the buffer)
  SE3_REQUIRE(x_rs >= 4 && x_rs % 4 == 0 && ((uintptr_t)x & 15) == 0 && rows * x_rs < (1ll << 29), SE3_ERR_INVALID_ARG,
              "linear_stream: x rows must be 16-byte aligned, row stride %lld, below 2 GB", (long long)x_rs);
  const bool transposed = t_rows > 0;
  if (transposed)
    SE3_REQUIRE(t_rows % 4 == 0 && rows % t_rows == 0 && t_ld >= t_rows && t_ld % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
                    (rows / t_rows) * N * t_ld < (1ll << 29),
                SE3_ERR_INVALID_ARG, "linear_stream: transposed output blocks of %d rows (a multiple of 4 dividing %lld), leading dimension %lld",
                t_rows, (long long)rows, (long long)t_ld);
  else
    SE3_REQUIRE(out_rs >= N && rows * out_rs < (1ll << 29), SE3_ERR_INVALID_ARG, "linear_stream: out row stride %lld", (long long)out_rs);
  if (rows == 0) return SE3_OK;
  // tile: 64 x 256 when that fills the chip, else 64 x 128 when THAT gives every compute unit a workgroup, else 64 x 64 (its weight ring is
  // four K-steps deep: -3 us on the K = 512 layers of a few thousand rows, -9 us at K = 1536; N <= 64: 128 x 64 / 128 x 32 as the unary layers)
  int TR, BN, cfg;
  const int64_t rt64 = se3_cdiv(rows, 64);
  if (N > 128) {
    if (rt64 * se3_cdiv(N, 256) >= 512) { TR = 64; BN = 256; cfg = 0; }
    else if (rt64 * se3_cdiv(N, 128) >= 320 || N % 128 > 64) { TR = 64; BN = 128; cfg = 1; }
    else { TR = 64; BN = 64; cfg = 2; }
  } else if (N > 64) { TR = rt64 >= 512 ? 128 : 64; BN = 128; cfg = rt64 >= 512 ? 3 : 1; }
  else if (N > 32) { TR = rt64 >= 512 ? 128 : 64; BN = 64; cfg = rt64 >= 512 ? 4 : 2; }
  else { TR = 128; BN = 32; cfg = 5; }
  if (transposed && cfg == 3) { TR = 64; cfg = 1; }          // (the 128 x 128 tile with the transposed epilogue spills)
  SegTable T;
  int ncb;
  const int chunks = dense_chunks(rows, N, nullptr, 1, T, TR, BN, ncb, true);
  SE3_REQUIRE(chunks > 0, SE3_ERR_INVALID_ARG, "linear_stream: %lld rows", (long long)rows);
  DenseArgs a{};
  a.x = x;
  a.K = K;
  a.N = N;
  a.NCT = (N + 63) / 64 * 2;
  a.hdr = static_cast<const float*>(weight_pieces);
  a.Wf = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces) + kHeaderB);
  a.out = out;
  a.xb = bias;
  a.T = T;
  a.final_slope = relu ? 0.f : 1.f;
  a.x_rs = (int)x_rs;
  a.out_rs = (int)out_rs;
  a.t_rows = t_rows;
  a.t_ld = (int)t_ld;
  if (seg_channels > 0) {
    SE3_REQUIRE(seg_channels % 32 == 0 && K % seg_channels == 0 && seg_stride >= rows * x_rs && (K / seg_channels) * seg_stride < (1ll << 29),
                SE3_ERR_INVALID_ARG, "linear_stream: %d input channels in segments of %d", K, seg_channels);
    a.seg_steps = seg_channels / 32;
    a.seg_stride = seg_stride;
  }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)chunks, (unsigned)ncb);
  switch (cfg) {
    case 0: plain_launch<1, 4, 2, 2>(a, transposed, grid, st); break;
    case 1: plain_launch<2, 2, 1, 2>(a, transposed, grid, st); break;
    case 2: plain_launch<2, 2, 1, 1>(a, transposed, grid, st); break;
    case 3: plain_launch<2, 2, 2, 2>(a, transposed, grid, st); break;
    case 4: plain_launch<4, 1, 1, 2>(a, transposed, grid, st); break;
    default: plain_launch<4, 1, 1, 1>(a, transposed, grid, st); break;
  }
  SE3_CHECK_LAUNCH("linear_stream");
  return SE3_OK;
}

// The kernel addresses its operands with 32-bit byte offsets from one base (buffer addressing: < 2 GB per operand).  Larger activations
// (more or larger clouds per forward: ADVICE round 4) are multiplied in row ranges of whole 256-row units (whole transposed blocks) that fit.
int linear_stream(const float* x, int64_t rows, int K, int64_t x_rs, const void* weight_pieces, const float* bias, int N, int relu, float* out,
                  int64_t out_rs, int t_rows, int64_t t_ld, void* stream, int seg_channels = 0, int64_t seg_stride = 0) {
  const int64_t lim = (1ll << 29) - 1;
  const bool transposed = t_rows > 0;
  const bool fits = rows >= 0 && x_rs > 0 && rows * x_rs <= lim &&
                    (transposed ? (rows / t_rows) * (int64_t)N * t_ld <= lim : rows * (out_rs > N ? out_rs : (int64_t)N) <= lim);
  if (fits || seg_channels > 0 || !x || !out || N <= 0) return linear_stream_one(x, rows, K, x_rs, weight_pieces, bias, N, relu, out, out_rs, t_rows, t_ld, stream, seg_channels, seg_stride);
  int64_t step;
  if (transposed) {
    SE3_REQUIRE(rows % t_rows == 0 && t_ld >= t_rows, SE3_ERR_INVALID_ARG, "linear_stream: transposed output blocks of %d rows", t_rows);
    const int64_t per_block = std::max((int64_t)t_rows * x_rs, (int64_t)N * t_ld);
    SE3_REQUIRE(per_block <= lim, SE3_ERR_UNSUPPORTED, "linear_stream: one transposed block of %d rows exceeds 2 GB", t_rows);
    step = lim / per_block * t_rows;
  } else {
    SE3_REQUIRE(out_rs >= N, SE3_ERR_INVALID_ARG, "linear_stream: out row stride %lld", (long long)out_rs);
    step = lim / std::max(x_rs, out_rs) / 256 * 256;
    SE3_REQUIRE(step > 0, SE3_ERR_UNSUPPORTED, "linear_stream: row strides %lld / %lld too large", (long long)x_rs, (long long)out_rs);
  }
  for (int64_t r0 = 0; r0 < rows; r0 += step) {
    const int64_t n = std::min(step, rows - r0);
    float* o = transposed ? out + (r0 / t_rows) * (int64_t)N * t_ld : out + r0 * out_rs;
    // (x_rs < K: the last row of a range reads zeros past the buffer bound where the single launch read the next row -- against zero weight columns)
    const int rc = linear_stream_one(x + r0 * x_rs, n, K, x_rs, weight_pieces, bias, N, relu, o, out_rs, t_rows, t_ld, stream);
    if (rc != SE3_OK) return rc;
  }
  return SE3_OK;
}
}  // namespace

extern "C" int se3_linear_stream(const float* x, int64_t rows, int in_features, int64_t x_row_stride, const void* weight_pieces, const float* bias,
                                 int out_features, int apply_relu, float* out, int64_t out_row_stride, void* stream) {
  return linear_stream(x, rows, in_features, x_row_stride, weight_pieces, bias, out_features, apply_relu, out, out_row_stride, 0, 0, stream);
}

// out_t (rows / block_rows, out_features, ld): the product of every block of `block_rows` consecutive rows stored transposed -- the value
// projection V^T (A, C, Rp) of the attention kernels from packed rows (A, R, C) in ONE launch (rpe_transformer.py:60-62 proj_v + the
// 'b n (h c) -> b h n c' rearrange; se3et_amd.functional.project_values_transposed).
extern "C" int se3_linear_stream_transposed(const float* x, int64_t rows, int in_features, int64_t x_row_stride, const void* weight_pieces,
                                            const float* bias, int out_features, int block_rows, float* out_t, int64_t ld, void* stream) {
  SE3_REQUIRE(block_rows > 0, SE3_ERR_INVALID_ARG, "linear_stream_transposed: block_rows %d", block_rows);
  return linear_stream(x, rows, in_features, x_row_stride, weight_pieces, bias, out_features, 0, out_t, 0, block_rows, ld, stream);
}


// x given as in_features / seg_channels arrays (rows, seg_channels) that lie seg_stride floats apart: row n of the product's input is the
// concatenation of row n of every array -- RotCompressOutput's 'b a n c -> b n (a c)' (output_layer.py:38-47) without the rearranged copy.
extern "C" int se3_linear_stream_segments(const float* x, int64_t rows, int in_features, int seg_channels, int64_t seg_stride,
                                          const void* weight_pieces, const float* bias, int out_features, int apply_relu, float* out,
                                          int64_t out_row_stride, void* stream) {
  SE3_REQUIRE(seg_channels > 0, SE3_ERR_INVALID_ARG, "linear_stream_segments: seg_channels %d", seg_channels);
  return linear_stream(x, rows, in_features, seg_channels, weight_pieces, bias, out_features, apply_relu, out, out_row_stride, 0, 0, stream,
                       seg_channels, seg_stride);
}
