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
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float f4get(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// =====================================================================================================================
// rpe_bias_kernel
// =====================================================================================================================
template <int CT, int RT, bool PREFETCH, int MINW>   // CT = C / 16 ; RT = row tiles of 16 folded queries (AH <= 16 RT)
__global__ __launch_bounds__(256, MINW) void rpe_bias_kernel(const float* __restrict__ qp, const float* __restrict__ qe,
                                                       int qp_rs, long long qp_sa, const float* __restrict__ emb,
                                                       const float* __restrict__ eq_emb, int N, int M, int AH, int H, int Mp,
                                                       float* __restrict__ bias) {
  constexpr int C = CT * 16;
  __shared__ float4 afrag[RT * CT * 64];
  __shared__ float4 qe_s[32];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // folded queries of row n: qp[a][n][h*C + c] (row stride qp_rs, anchor stride qp_sa) -> LDS in MFMA-fragment order
  // afrag[(rt*CT + t)*64 + kq*16 + r] = float4 of channels 16 t + 4 kq .. +3 of folded query row 16 rt + r (r = a*H + h).
  // Thread i takes row (i % (16 RT)) and chunk i / (16 RT): consecutive lanes write consecutive float4 (conflict free).
  for (int i = threadIdx.x; i < RT * 16 * (C / 4); i += 256) {
    const int row = i % (RT * 16), c4 = i / (RT * 16);
    const int t = c4 >> 2, kq = c4 & 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < AH) {
      const int a = row / H, h = row - a * H;
      v = ld4(qp + (size_t)a * qp_sa + (size_t)n * qp_rs + h * C + 4 * c4);
    }
    afrag[((row >> 4) * CT + t) * 64 + kq * 16 + (row & 15)] = v;
  }
  if (threadIdx.x < 32) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qe != nullptr && threadIdx.x < AH) {
      const int a = threadIdx.x / H, h = threadIdx.x - a * H;
      v = ld4(qe + (size_t)a * qp_sa + (size_t)n * qp_rs + 4 * h);
    }
    qe_s[threadIdx.x] = v;
  }
  __syncthreads();

  const int tiles = (M + 15) >> 4;
  const int per = (tiles + gridDim.y - 1) / gridDim.y;
  const int t_begin = blockIdx.y * per, t_end = min(tiles, t_begin + per);
  const int col = lane & 15, kq = lane >> 4;
  const float* Erow0 = emb + (size_t)n * M * C + 4 * kq;
  const bool uniform_anchor = (H % 4) == 0;      // the 4 rows a lane owns in a row tile then share one anchor

  int tile = t_begin + wave;
  float4 b[CT], bn[PREFETCH ? CT : 1];
  if (PREFETCH && tile < t_end) {
    const float* Er = Erow0 + (size_t)min((tile << 4) + col, M - 1) * C;
#pragma unroll
    for (int t = 0; t < CT; t++) b[t] = ld4(Er + 16 * t);
  }
  for (; tile < t_end; tile += 4) {
    // compiler fence: without it the loop-invariant LDS fragment reads are hoisted out of the tile loop (128 VGPRs) and
    // spilled to scratch -- the reads must stay inside the loop, next to their MFMAs
    asm volatile("" ::: "memory");
    const int m0 = tile << 4;
    const bool more = PREFETCH && tile + 4 < t_end;
    if (PREFETCH) {
      if (more) {                                  // software prefetch of the next tile (second register set)
        const float* Er = Erow0 + (size_t)min(((tile + 4) << 4) + col, M - 1) * C;
#pragma unroll
        for (int t = 0; t < CT; t++) bn[t] = ld4(Er + 16 * t);
      }
    } else {
      const float* Er = Erow0 + (size_t)min(m0 + col, M - 1) * C;
#pragma unroll
      for (int t = 0; t < CT; t++) b[t] = ld4(Er + 16 * t);
    }
    const int m = m0 + col;
    float4 e4[RT];
    if (eq_emb != nullptr && uniform_anchor) {
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        const int a = min((16 * rt + 4 * kq) / H, AH / H - 1);
        e4[rt] = ld4(eq_emb + (((size_t)a * N + n) * M + min(m, M - 1)) * 4);
      }
    }
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < CT; t++) {
      float4 a[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) a[rt] = afrag[(rt * CT + t) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(a[rt], i), f4get(b[t], i), acc[rt], 0, 0, 0);
      }
    }
    if (m < M) {
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int row = 16 * rt + 4 * kq + j;
          if (row < AH) {
            float val = acc[rt][j];
            if (eq_emb != nullptr) {
              const float4 e = uniform_anchor ? e4[rt] : ld4(eq_emb + (((size_t)(row / H) * N + n) * M + m) * 4);
              const float4 w = qe_s[row];
              val += (w.x * e.x + w.y * e.y) + (w.z * e.z + w.w * e.w);
            }
            bias[((size_t)row * N + n) * Mp + m] = val;
          }
        }
      }
    }
    if (PREFETCH && more) {
#pragma unroll
      for (int t = 0; t < CT; t++) b[t] = bn[t];
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
// The loads of tile t+step are issued before the MFMAs of tile t (two register sets).
template <int D>
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
  if (tile < tiles) flash_load<D>(cur, k, v, bias_row, tile << 5, M, k_rs, v_rs);
  for (; tile < tiles; tile += tile_step) {
    const int m0 = tile << 5;
    const bool more = tile + tile_step < tiles;
    if (more) flash_load<D>(nxt, k, v, bias_row, (tile + tile_step) << 5, M, k_rs, v_rs);
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
    if (more) cur = nxt;
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
  int N, M, C, H, Mp;
  int q_rs, k_rs, v_rs;               // row strides of q, k (>= C) and of the transposed values (>= ceil32(M)) in floats
  long long q_sa, k_sa, v_sa, o_sa;   // anchor strides in floats (0 = shared by all anchors)
  float scale;
};

// grid (ceil(N/32), H, A_out); NW waves split the key tiles of one (anchor, head, 32-query tile)
template <int D, int NW, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void attention_kernel(AttnArgs p) {
  __shared__ float sm[64 * NW], sl[64 * NW];
  __shared__ float so[NW * FlashState<D>::DT * 16 * 64];
  const int n0 = blockIdx.x * 32, h = blockIdx.y, a = blockIdx.z;
  const int wave = threadIdx.x >> 6;
  FlashState<D> st;
  flash_init(st);
  const float* bias = p.bias ? p.bias + ((size_t)(a * p.H + h) * p.N) * p.Mp : nullptr;
  flash_tiles<D>(st, p.q + a * p.q_sa + h * D, p.k + a * p.k_sa + h * D, p.v + a * p.v_sa + (size_t)h * D * p.v_rs, bias, n0, p.N, p.M, p.q_rs,
                 p.k_rs, p.v_rs, p.Mp, p.scale, wave, NW);
  flash_merge<D, NW>(st, sm, sl, so);
  if (wave == 0) flash_store<D>(st, p.out + a * p.o_sa + h * D, n0, p.N, p.C, 1.f, false);
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

// one workgroup: g[a,e] = sum_i partial[(a*A+e)*P + i] / (N M) -> mixing weights.
//   mode 0 (a_soft): mix[a,e] = g[a,e] / sum_e g[a,e];  weights = mix (A*A values)
//   mode 1 (r_soft): w[r] = mean_a g[a, trace[r,a]] normalised over r; mix[a,e] = sum_{r: trace[r,a]=e} w[r]; weights = w (R values)
__global__ __launch_bounds__(64) void cross_eq_mix_kernel(const float* __restrict__ partial, int P, float inv_nm, int A, int R,
                                                          const int64_t* __restrict__ trace, int mode, float* __restrict__ mix,
                                                          float* __restrict__ weights) {
  __shared__ float g[64], w[64], tot;
  const int t = threadIdx.x;
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

static int g_attn_variant = 0;
extern "C" void se3_debug_set_attention_variant(int variant) { g_attn_variant = variant; }
static int g_bias_variant = 0;
static int g_bias_split = 0;
// tuning hooks (benchmarks only): kernel variant / m-split override; 0 = default
extern "C" void se3_debug_set_bias_variant(int variant, int split) { g_bias_variant = variant; g_bias_split = split; }

extern "C" int se3_rpe_bias_fwd(const float* qp, const float* qe, int row_stride, int64_t anchor_stride, const float* emb,
                                const float* eq_emb, int N, int M, int C, int AH, int H, int bias_row_stride, float* bias,
                                void* stream) {
  SE3_REQUIRE(qp && emb && bias, SE3_ERR_INVALID_ARG, "rpe_bias: null pointer");
  SE3_REQUIRE(row_stride % 4 == 0 && row_stride >= H * C, SE3_ERR_INVALID_ARG, "rpe_bias: folded-query row stride");
  SE3_REQUIRE((qe == nullptr) == (eq_emb == nullptr), SE3_ERR_INVALID_ARG, "rpe_bias: qe and eq_emb go together");
  SE3_REQUIRE(N >= 1 && M >= 1 && AH >= 1 && AH <= 32 && H >= 1 && AH % H == 0, SE3_ERR_UNSUPPORTED,
              "rpe_bias: N %d M %d AH %d H %d", N, M, AH, H);
  SE3_REQUIRE(bias_row_stride >= M, SE3_ERR_INVALID_ARG, "rpe_bias: bias row stride < M");
  hipStream_t st = (hipStream_t)stream;
  int split = (768 + N - 1) / N;
  const int tiles = (M + 15) / 16;
  if (split > (tiles + 3) / 4) split = (tiles + 3) / 4;
  if (split < 1) split = 1;
  if (g_bias_split > 0) split = g_bias_split;
  dim3 grid((unsigned)N, (unsigned)split);
#define SE3_BIAS_ARGS qp, qe, row_stride, anchor_stride, emb, eq_emb, N, M, AH, H, bias_row_stride, bias
#define SE3_BIAS_LAUNCH(CT)                                                                                          \
  if (AH <= 16) {                                                                                                    \
    switch (g_bias_variant) {                                                                                        \
      case 1: rpe_bias_kernel<CT, 1, true, 2><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                           \
      case 2: rpe_bias_kernel<CT, 1, true, 3><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                           \
      case 3: rpe_bias_kernel<CT, 1, false, 3><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                          \
      case 4: rpe_bias_kernel<CT, 1, false, 5><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                          \
      default: rpe_bias_kernel<CT, 1, false, 4><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                         \
    }                                                                                                                \
  } else {                                                                                                           \
    switch (g_bias_variant) {                                                                                        \
      case 1: rpe_bias_kernel<CT, 2, true, 2><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                           \
      case 2: rpe_bias_kernel<CT, 2, true, 3><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                           \
      case 3: rpe_bias_kernel<CT, 2, false, 3><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                          \
      case 4: rpe_bias_kernel<CT, 2, false, 5><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                          \
      default: rpe_bias_kernel<CT, 2, false, 4><<<grid, 256, 0, st>>>(SE3_BIAS_ARGS); break;                         \
    }                                                                                                                \
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
  SE3_CHECK_LAUNCH("rpe_bias");
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
  AttnArgs p{q, k, v, bias, out, N, M, C, H, bias_row_stride, q_row_stride, k_row_stride, v_row_stride, q_anchor_stride, k_anchor_stride,
             v_anchor_stride, out_anchor_stride, scale};
  dim3 grid((unsigned)((N + 31) / 32), (unsigned)H, (unsigned)num_anchors);
  hipStream_t st = (hipStream_t)stream;
  const bool wide = (M + 31) / 32 >= 6;      // enough key tiles to feed 8 waves per (anchor, head, query tile)
  int rc = dispatch_head_dim(C / H, [&](auto d) {
    constexpr int D = decltype(d)::value;
    (void)wide;
    switch (g_attn_variant) {
      case 1: attention_kernel<D, 3, 1><<<grid, 192, 0, st>>>(p); break;
      case 2: attention_kernel<D, 4, 1><<<grid, 256, 0, st>>>(p); break;
      case 3: attention_kernel<D, 2, 1><<<grid, 128, 0, st>>>(p); break;
      case 4: attention_kernel<D, 6, 1><<<grid, 384, 0, st>>>(p); break;
      default: attention_kernel<D, 4, 2><<<grid, 256, 0, st>>>(p); break;
    }
  }, "attention");
  if (rc != SE3_OK) return rc;
  SE3_CHECK_LAUNCH("attention");
  return SE3_OK;
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
                                                       num_rotations, trace_idx, mode, mix, weights);
  SE3_CHECK_LAUNCH("cross_eq_mix");
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
