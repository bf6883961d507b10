// G1/G2: geometric structure embedding of the SE3ET coarse transformer.
//
// Reference (geotransformer/modules/geotransformer/geotransformer.py:57-121, transformer/positional_embedding.py:8-34):
//   d_idx[n,m]   = sqrt(clamp(|p_n|^2 - 2 p_n.p_m + |p_m|^2, 0)) / sigma_d
//   a_idx[n,m,k] = atan2(|r_k x v|, r_k . v) * 180 / (sigma_a pi),  r_k = p_knn(n,k) - p_n (k = 3 nearest, self excluded),
//                                                                    v = p_m - p_n
//   E[n,m,:]     = W_d emb(d_idx) + b_d + max_k (W_a emb(a_idx_k) + b_a),  emb(x)[2i] = sin(x w_i), emb(x)[2i+1] = cos(x w_i)
//   Eeq[a,n,m,:] = [Y0, R_a^T Y1(p_n - p_m)]   (l <= 1 real spherical harmonics; convention: DESIGN.md section 2)
// The reference materialises the (N, N, 3, C) sinusoid tensors (897 MB at N = 382) and runs an 8 N^2 C^2 = 76 GF GEMM.
// Here f_d(x) = W_d emb(x) + b_d and f_a(x) = W_a emb(x) + b_a are treated as what they are -- smooth vector-valued functions
// of ONE scalar (sums of 128 sinusoids with angular frequency <= 1 per index unit): the host tabulates (f, f') on a uniform
// grid with two tiny library GEMMs (cached per weight version) and this kernel evaluates them by cubic Hermite interpolation
// (error h^4/384 max|f''''|, ~1e-9 at h = 1/64), reading the L2-resident tables instead of doing 2 C^2 flops per sample.
// Indices beyond the table are evaluated with the exact sinusoid sum (slow path, wave-uniform branch).
//
// Mapping: one workgroup per (n, block of m); thread c owns channel c (table rows are read fully coalesced).
#include "common.h"
#include "geo_records.h"

namespace {

using se3geo::EmbParams;
using se3geo::hermite_weights;

__device__ __forceinline__ float hermite(const float2* __restrict__ tab, int C, int c, float x, float inv_h, int entries,
                                         bool& ok) {
  const float u = x * inv_h;
  int j = (int)floorf(u);
  ok = (j >= 0) && (j + 1 < entries);
  j = min(max(j, 0), entries - 2);
  const float t = u - (float)j;
  const float2 p0 = tab[(size_t)j * C + c], p1 = tab[(size_t)(j + 1) * C + c];
  const float4 w = hermite_weights(t, 1.0f / inv_h);
  return (w.x * p0.x + w.y * p1.x) + (w.z * p0.y + w.w * p1.y);
}

// exact evaluation W[c, :] . emb(x) + b[c] (fallback for indices outside the table)
__device__ __attribute__((noinline)) float exact_eval(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ div_term,
                            int C, int c, float x) {
  float acc = b[c];
  for (int i = 0; i < C / 2; i++) {
    float s, co;
    sincosf(x * div_term[i], &s, &co);
    acc += W[(size_t)c * C + 2 * i] * s + W[(size_t)c * C + 2 * i + 1] * co;
  }
  return acc;
}

constexpr int kMB = 16;    // m values per workgroup iteration (indices and interpolation weights staged in LDS)

// Per (m, term) the table interval and the four Hermite weights are the same for all C channels: they are computed once by
// the staging threads (4 terms x kMB values) and broadcast from LDS, which leaves two table reads and four FMAs per channel and
// term in the channel loop (the kernel is VALU bound: ~150 instructions per output before, ~50 now).
__global__ void geo_embedding_kernel(const float* __restrict__ pts, const int64_t* __restrict__ knn, int N, int C,
                                     const float2* __restrict__ tab_d, const float2* __restrict__ tab_a, EmbParams P,
                                     const float* __restrict__ Wd, const float* __restrict__ bd,
                                     const float* __restrict__ Wa, const float* __restrict__ ba,
                                     const float* __restrict__ div_term, const float* __restrict__ wigner_d1,
                                     void* __restrict__ emb_out, int emb_bf16, float* __restrict__ eq_emb, int A) {
  float* emb = static_cast<float*>(emb_out);
  unsigned short* emb16 = static_cast<unsigned short*>(emb_out);
  __shared__ float idx_s[kMB][4];
  __shared__ float4 wt_s[kMB][4];         // (h00, h01, h h10, h h11) of term t (0 = distance, 1..3 = angles)
  __shared__ int j_s[kMB][4];             // table interval, -1 = outside the table (exact evaluation)
  __shared__ float unit_s[kMB][3];
  const int n = blockIdx.x;
  const int m_per = (N + gridDim.y - 1) / gridDim.y;
  const int m_begin = blockIdx.y * m_per, m_end = min(N, m_begin + m_per);
  const float px = pts[3 * n], py = pts[3 * n + 1], pz = pts[3 * n + 2];
  float rx[3], ry[3], rz[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int64_t j = knn[3 * n + k];
    rx[k] = pts[3 * j] - px; ry[k] = pts[3 * j + 1] - py; rz[k] = pts[3 * j + 2] - pz;
  }
  const float nn2 = se3_ref_sq_norm(px, py, pz);
  for (int m0 = m_begin; m0 < m_end; m0 += kMB) {
    __syncthreads();
    if (threadIdx.x < kMB) {
      const int m = min(m0 + (int)threadIdx.x, N - 1);
      const float qx = pts[3 * m], qy = pts[3 * m + 1], qz = pts[3 * m + 2];
      const float d2 = se3_ref_sq_dist(px, py, pz, nn2, qx, qy, qz, se3_ref_sq_norm(qx, qy, qz));
      float x[4];
      x[0] = sqrtf(d2) * P.sigma_d_inv;
      const float vx = qx - px, vy = qy - py, vz = qz - pz;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float cx = ry[k] * vz - rz[k] * vy, cy = rz[k] * vx - rx[k] * vz, cz = rx[k] * vy - ry[k] * vx;
        const float sn = sqrtf(cx * cx + cy * cy + cz * cz);
        float cs = rx[k] * vx + ry[k] * vy + rz[k] * vz;
        cs = (cs == 0.f) ? 0.f : cs;      // torch.sum yields +0 for an all-(-0) sum (the n == m diagonal): atan2(0, +0) = 0, not pi
        x[1 + k] = atan2f(sn, cs) * P.factor_a;
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const float inv_h = t == 0 ? P.d_inv_h : P.a_inv_h;
        const int entries = t == 0 ? P.d_entries : P.a_entries;
        const float u = x[t] * inv_h;
        const int j = (int)floorf(u);
        const bool ok = (j >= 0) && (j + 1 < entries);
        const float tt = u - (float)j, h = 1.0f / inv_h;
        idx_s[threadIdx.x][t] = x[t];
        j_s[threadIdx.x][t] = ok ? j : -1;
        wt_s[threadIdx.x][t] = hermite_weights(tt, h);
      }
      // unit vector of p_n - p_m for the equivariant embedding (zero vector -> 0, as F.normalize with eps 1e-12)
      const float len = sqrtf(vx * vx + vy * vy + vz * vz);
      const float inv = 1.f / fmaxf(len, 1e-12f);
      unit_s[threadIdx.x][0] = -vx * inv; unit_s[threadIdx.x][1] = -vy * inv; unit_s[threadIdx.x][2] = -vz * inv;
    }
    __syncthreads();
    const int cnt = min(kMB, m_end - m0);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      for (int i = 0; i < cnt; i++) {
        float val[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int j = j_s[i][t];
          if (j >= 0) {                                     // wave-uniform
            const float2* tab = t == 0 ? tab_d : tab_a;
            const float2 p0 = tab[(size_t)j * C + c], p1 = tab[(size_t)(j + 1) * C + c];
            const float4 w = wt_s[i][t];
            val[t] = (w.x * p0.x + w.y * p1.x) + (w.z * p0.y + w.w * p1.y);
          } else {
            val[t] = t == 0 ? exact_eval(Wd, bd, div_term, C, c, idx_s[i][0]) : exact_eval(Wa, ba, div_term, C, c, idx_s[i][t]);
          }
        }
        const float e = val[0] + fmaxf(fmaxf(val[1], val[2]), val[3]);
        if (emb_bf16) {                                       // round to nearest even (values are finite)
          unsigned u = __float_as_uint(e);
          u += 0x7fffu + ((u >> 16) & 1u);
          __builtin_nontemporal_store((unsigned short)(u >> 16), &emb16[((size_t)n * N + (m0 + i)) * C + c]);
        } else {
          __builtin_nontemporal_store(e, &emb[((size_t)n * N + (m0 + i)) * C + c]);
        }
      }
    }
    if (eq_emb != nullptr) {
      for (int e = threadIdx.x; e < cnt * A; e += blockDim.x) {
        const int i = e / A, a = e - i * A;
        const float ux = unit_s[i][0], uy = unit_s[i][1], uz = unit_s[i][2];
        const float* D = wigner_d1 + 9 * a;          // D^1_a (3, 3): out_c = sum_d D[c][d] Y1_d
        const float c1 = 0.4886025119029199f;        // sqrt(3 / (4 pi))
        float4 o;
        o.x = 0.28209479177387814f;                  // 1 / (2 sqrt(pi))
        o.y = c1 * (D[0] * ux + D[1] * uy + D[2] * uz);
        o.z = c1 * (D[3] * ux + D[4] * uy + D[5] * uz);
        o.w = c1 * (D[6] * ux + D[7] * uy + D[8] * uz);
        reinterpret_cast<float4*>(eq_emb)[((size_t)a * N + n) * N + (m0 + i)] = o;
      }
    }
  }
}

// ---- channel-slice form: the angle table of the slice lives in LDS ----------------------------------------------------------------
// The kernel above reads 64 B of table per 4 B written, all of it from L2 (the tables are far larger than L1): at N = 358 that is
// 2.1 GB per cloud at the ~15 TB/s the chip gathers from L2 -- 140 us, 0.95 TB/s of output.  Three of the four terms read the ANGLE table
// (index range 180 / sigma_a + 1 units).  Here the work is split in two kernels:
//   geo_pair_terms_kernel   one thread per (n, m): the index (distance / one of the three angles), its table interval and the four
//                           Hermite weights -> a (N N, 4) record array (80 B per pair, L2 / Infinity-Cache resident); also writes Eeq.
//   geo_embedding_slice_kernel   a workgroup owns kCS = 32 channels and a block of query rows and keeps its slice of the angle table
//                           (entries x 256 B, 105 KB at 32 entries per unit) in LDS: the three angle terms cost 96 B of ds_read_b128 per
//                           lane and pair, only the distance term still comes from L2 (16 B per output).  A lane owns two channels (one
//                           float4 = (f, f') x 2 per table entry), 16 lanes one pair: a wave writes four full 128-byte lines per store.
//                           A wave owns a contiguous run of pairs; their records travel global -> registers -> a wave-private LDS ring
//                           in coalesced blocks of 16 pairs, two blocks ahead of their use, and are read back as 16-lane broadcasts.
//                           No barrier after the table fill: the 16 waves of the one workgroup per CU run free.
// Measured on the way (N = 358, us per cloud; the single-kernel form: 167): records staged in LDS behind a barrier per 64 pairs, 8 waves:
// 174; records as per-lane vector loads (five broadcast 16-byte loads per pair cost the L1 return path as much as 5 KB of real data per
// wave): 97; one pair per wave with the record in SGPRs through the scalar cache (every record a scalar-cache miss; the SGPR file holds
// too few records to cover that latency): 96-113.
constexpr int kCS = 32;      // channels per workgroup
constexpr int kSliceThreads = 1024;
constexpr int kRecBlock = 16;                                  // pairs per staged record block
constexpr int kRecBytes = kRecBlock * (64 + 16);               // weights (4 float4) + intervals (int4) per pair

struct ExactArgs { const float *Wd, *bd, *Wa, *ba, *div_term; };    // only the cold path (index outside a table) reads these

template <bool BF16>
__global__ __launch_bounds__(kSliceThreads) void geo_embedding_slice_kernel(
    int N, int C, const float2* __restrict__ tab_d, const float2* __restrict__ tab_a, int a_entries, const int4* __restrict__ jrec,
    const float4* __restrict__ wrec, ExactArgs X, void* __restrict__ emb_out, int rows_per_block) {
  extern __shared__ __align__(16) float4 lds4[];
  float4* atab = lds4 + (kSliceThreads / 64) * 2 * (kRecBytes / 16);                  // [a_entries][16]: (f, f, f', f') of channels 2 l, 2 l + 1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4* ring = lds4 + wave * 2 * (kRecBytes / 16);                                  // [2][64 weights | 16 intervals] of this wave
  const int c_base = blockIdx.x * kCS;
  const int n0 = blockIdx.y * rows_per_block, n1 = min(N, n0 + rows_per_block);
  if (n0 >= n1) return;
  for (int i = tid; i < a_entries * 16; i += kSliceThreads) {
    const int e = i >> 4, l = i & 15;
    const float4 t = *reinterpret_cast<const float4*>(tab_a + (size_t)e * C + c_base + 2 * l);
    atab[i] = make_float4(t.x, t.z, t.y, t.w);               // (f, f) (f', f') of the two channels: operand pairs of the packed FMAs
  }
  __syncthreads();
  const unsigned first = (unsigned)n0 * (unsigned)N, last = (unsigned)n1 * (unsigned)N;   // the block's pairs: contiguous (n, m) records
  float* emb = static_cast<float*>(emb_out);
  unsigned* emb16 = static_cast<unsigned*>(emb_out);
  constexpr unsigned kWaves = kSliceThreads / 64;
  const unsigned chunk = ((last - first + kWaves - 1) / kWaves + (kRecBlock - 1)) & ~(unsigned)(kRecBlock - 1);
  const unsigned wbeg = first + (unsigned)wave * chunk;
  if (wbeg >= last) return;
  const unsigned wend = min(last, wbeg + chunk);
  const int blocks = (int)((wend - wbeg + kRecBlock - 1) / kRecBlock);
  const int l = lane & 15, grp = lane >> 4, c0 = c_base + 2 * l;
  const float2* dcol = tab_d + c0;
  // record block b of this wave: lane -> one float4 of the 64 weights, lanes (l, any group) -> interval record l.  Slots past the end of
  // the run hold the run's LAST pair: those lanes recompute and re-store that pair's values, which keeps every store unconditional (a
  // store inside a branch makes the number of outstanding vector-memory operations path dependent, and the compiler then drains them
  // all -- the write acknowledgements included -- in front of every use of a prefetched load)
#define SE3_REQUEST(b_)                                                                                       \
  rw = wrec[min(wbeg + (unsigned)(b_) * kRecBlock + (unsigned)(lane >> 2), wend - 1) * 4 + (unsigned)(lane & 3)]; \
  rj = jrec[min(wbeg + (unsigned)(b_) * kRecBlock + (unsigned)l, wend - 1)];
#define SE3_PUBLISH(b_)                                                                      \
  {                                                                                          \
    float4* dst = ring + ((b_) & 1) * (kRecBytes / 16);                                      \
    dst[lane] = rw;                                                                          \
    if (grp == 0) reinterpret_cast<int4*>(dst + 64)[l] = rj;                                 \
  }
  float4 rw;
  int4 rj;
  SE3_REQUEST(0)
  SE3_PUBLISH(0)
  // pipeline state: intervals and distance rows of the iteration about to be evaluated
  int4 jc = reinterpret_cast<const int4*>(ring + 64)[grp];
  float4 dc0, dc1;
  {
    const int jd = max(jc.x, 0);
    dc0 = *reinterpret_cast<const float4*>(dcol + (size_t)jd * C);
    dc1 = *reinterpret_cast<const float4*>(dcol + (size_t)(jd + 1) * C);
  }
  for (int blk = 0; blk < blocks; blk++) {
    SE3_REQUEST(blk + 1)                                        // past the run: clamped, never used
#pragma unroll
    for (int sub = 0; sub < 4; sub++) {
      if (sub == 3) SE3_PUBLISH(blk + 1)                        // the next iteration's intervals are read just below
      const float4* cur = ring + (blk & 1) * (kRecBytes / 16);
      // intervals of the next iteration -> its distance rows (L2) are requested one iteration ahead
      const float4* nxt = sub == 3 ? ring + ((blk + 1) & 1) * (kRecBytes / 16) : cur;
      const int4 jn = reinterpret_cast<const int4*>(nxt + 64)[((sub + 1) & 3) * 4 + grp];
      const int jdn = max(jn.x, 0);
      const float4 dn0 = *reinterpret_cast<const float4*>(dcol + (unsigned)jdn * (unsigned)C);      // 32-bit offsets (table < 4 GB)
      const float4 dn1 = *reinterpret_cast<const float4*>(dcol + (unsigned)(jdn + 1) * (unsigned)C);
      __builtin_amdgcn_sched_barrier(0);      // the store below stays YOUNGER than these loads: waiting for them must not wait for its acknowledgement
      // this iteration: pair = wbeg + 16 blk + 4 sub + grp
      const unsigned pair = min(wbeg + (unsigned)(blk * kRecBlock + sub * 4 + grp), wend - 1);
      const float4* wp = cur + (sub * 4 + grp) * 4;
      const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
      const int j1 = max(jc.y, 0), j2 = max(jc.z, 0), j3 = max(jc.w, 0);
      const float4 a10 = atab[j1 * 16 + l], a11 = atab[(j1 + 1) * 16 + l];
      const float4 a20 = atab[j2 * 16 + l], a21 = atab[(j2 + 1) * 16 + l];
      const float4 a30 = atab[j3 * 16 + l], a31 = atab[(j3 + 1) * 16 + l];
      // explicit roundings (no contraction): the f32 and bf16 instantiations must produce the same f32 value
#define SE3_HERMITE(w_, f0_, g0_, f1_, g1_) \
  __fadd_rn(__fmaf_rn((w_).x, f0_, __fmul_rn((w_).y, f1_)), __fmaf_rn((w_).z, g0_, __fmul_rn((w_).w, g1_)))
      float x0 = SE3_HERMITE(w0, dc0.x, dc0.y, dc1.x, dc1.y), y0 = SE3_HERMITE(w0, dc0.z, dc0.w, dc1.z, dc1.w);
      float x1 = SE3_HERMITE(w1, a10.x, a10.z, a11.x, a11.z), y1 = SE3_HERMITE(w1, a10.y, a10.w, a11.y, a11.w);
      float x2 = SE3_HERMITE(w2, a20.x, a20.z, a21.x, a21.z), y2 = SE3_HERMITE(w2, a20.y, a20.w, a21.y, a21.w);
      float x3 = SE3_HERMITE(w3, a30.x, a30.z, a31.x, a31.z), y3 = SE3_HERMITE(w3, a30.y, a30.w, a31.y, a31.w);
#undef SE3_HERMITE
      if ((jc.x | jc.y | jc.z | jc.w) < 0) {                   // cold: an index outside its table (the record then holds the index itself)
        if (jc.x < 0) { x0 = exact_eval(X.Wd, X.bd, X.div_term, C, c0, w0.x); y0 = exact_eval(X.Wd, X.bd, X.div_term, C, c0 + 1, w0.x); }
        if (jc.y < 0) { x1 = exact_eval(X.Wa, X.ba, X.div_term, C, c0, w1.x); y1 = exact_eval(X.Wa, X.ba, X.div_term, C, c0 + 1, w1.x); }
        if (jc.z < 0) { x2 = exact_eval(X.Wa, X.ba, X.div_term, C, c0, w2.x); y2 = exact_eval(X.Wa, X.ba, X.div_term, C, c0 + 1, w2.x); }
        if (jc.w < 0) { x3 = exact_eval(X.Wa, X.ba, X.div_term, C, c0, w3.x); y3 = exact_eval(X.Wa, X.ba, X.div_term, C, c0 + 1, w3.x); }
      }
      const float ex = __fadd_rn(x0, fmaxf(fmaxf(x1, x2), x3));
      const float ey = __fadd_rn(y0, fmaxf(fmaxf(y1, y2), y3));
      const size_t o = (size_t)pair * C + c0;
      if (BF16) {                                              // round to nearest even (values are finite)
        unsigned ux = __float_as_uint(ex), uy = __float_as_uint(ey);
        ux += 0x7fffu + ((ux >> 16) & 1u);
        uy += 0x7fffu + ((uy >> 16) & 1u);
        __builtin_nontemporal_store((ux >> 16) | (uy & 0xffff0000u), &emb16[o >> 1]);
      } else {
        float2 e2 = make_float2(ex, ey);
        __builtin_nontemporal_store(*reinterpret_cast<unsigned long long*>(&e2), reinterpret_cast<unsigned long long*>(emb + o));
      }
      jc = jn; dc0 = dn0; dc1 = dn1;
    }
  }
#undef SE3_REQUEST
#undef SE3_PUBLISH
}

// ---- backward operands (training step) ---------------------------------------------------------------------------------------------------
// dL/dW_d = dE^T emb(d),  dL/dW_a = sum_k (dE masked to the channels where angle k holds the maximum)^T emb(angle_k),  dL/db_d = dL/db_a =
// sum dE: the products are library GEMMs; this kernel writes their operands in one pass over the pairs -- the sinusoid embeddings of the
// four indices S (4, N N, C) and the masked gradients dEk (3, N N, C).  The arg-max is taken on the same tabulated responses as the
// forward pass (first index on ties: tied angles of one pair are equal angles, e.g. the n == m diagonal, and then the split is immaterial).
__global__ void geo_embedding_bwd_operands_kernel(const float* __restrict__ pts, const int64_t* __restrict__ knn, int N, int C,
                                                  const float2* __restrict__ tab_a, EmbParams P, const float* __restrict__ Wa,
                                                  const float* __restrict__ ba, const float* __restrict__ div_term,
                                                  const float* __restrict__ dE, float* __restrict__ S, float* __restrict__ dEk) {
  __shared__ float idx_s[kMB][4];
  __shared__ float4 wt_s[kMB][4];
  __shared__ int j_s[kMB][4];
  const int n = blockIdx.x;
  const int m_per = (N + gridDim.y - 1) / gridDim.y;
  const int m_begin = blockIdx.y * m_per, m_end = min(N, m_begin + m_per);
  const float px = pts[3 * n], py = pts[3 * n + 1], pz = pts[3 * n + 2];
  float rx[3], ry[3], rz[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int64_t j = knn[3 * n + k];
    rx[k] = pts[3 * j] - px; ry[k] = pts[3 * j + 1] - py; rz[k] = pts[3 * j + 2] - pz;
  }
  const float nn2 = se3_ref_sq_norm(px, py, pz);
  const size_t plane = (size_t)N * N * C;
  for (int m0 = m_begin; m0 < m_end; m0 += kMB) {
    __syncthreads();
    if (threadIdx.x < kMB) {
      const int m = min(m0 + (int)threadIdx.x, N - 1);
      const float qx = pts[3 * m], qy = pts[3 * m + 1], qz = pts[3 * m + 2];
      const float d2 = se3_ref_sq_dist(px, py, pz, nn2, qx, qy, qz, se3_ref_sq_norm(qx, qy, qz));
      float x[4];
      x[0] = sqrtf(d2) * P.sigma_d_inv;
      const float vx = qx - px, vy = qy - py, vz = qz - pz;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float cx = ry[k] * vz - rz[k] * vy, cy = rz[k] * vx - rx[k] * vz, cz = rx[k] * vy - ry[k] * vx;
        const float sn = sqrtf(cx * cx + cy * cy + cz * cz);
        float cs = rx[k] * vx + ry[k] * vy + rz[k] * vz;
        cs = (cs == 0.f) ? 0.f : cs;
        x[1 + k] = atan2f(sn, cs) * P.factor_a;
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        idx_s[threadIdx.x][t] = x[t];
        if (t > 0) {
          const float u = x[t] * P.a_inv_h;
          const int j = (int)floorf(u);
          const bool ok = (j >= 0) && (j + 1 < P.a_entries);
          const float tt = u - (float)j, h = 1.0f / P.a_inv_h;
            j_s[threadIdx.x][t] = ok ? j : -1;
          wt_s[threadIdx.x][t] = hermite_weights(tt, h);
        }
      }
    }
    __syncthreads();
    const int cnt = min(kMB, m_end - m0);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float om = div_term[c >> 1];
      for (int i = 0; i < cnt; i++) {
        float val[3];
#pragma unroll
        for (int t = 1; t < 4; t++) {
          const int j = j_s[i][t];
          if (j >= 0) {
            const float2 p0 = tab_a[(size_t)j * C + c], p1 = tab_a[(size_t)(j + 1) * C + c];
            const float4 w = wt_s[i][t];
            val[t - 1] = (w.x * p0.x + w.y * p1.x) + (w.z * p0.y + w.w * p1.y);
          } else {
            val[t - 1] = exact_eval(Wa, ba, div_term, C, c, idx_s[i][t]);
          }
        }
        const int kb = val[1] > val[0] ? (val[2] > val[1] ? 2 : 1) : (val[2] > val[0] ? 2 : 0);
        const size_t o = ((size_t)n * N + (m0 + i)) * C + c;
        const float g = dE[o];
#pragma unroll
        for (int k = 0; k < 3; k++) dEk[k * plane + o] = k == kb ? g : 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          float sn, cs;
          sincosf(idx_s[i][t] * om, &sn, &cs);
          S[t * plane + o] = (c & 1) ? cs : sn;
        }
      }
    }
  }
}

// ---- table construction on the device -------------------------------------------------------------------------------
// tab (entries, C, 2) = (f, f') of f(x) = W emb(x) + b at x = j / per_unit, accumulated in float64.  Two launches in front of every
// embedding call validate the table against the CURRENT weights by content: 64 workgroups hash disjoint slices of (W, b, div_term)
// (position-keyed mixing, summed: independent of the slicing), then every workgroup of the build kernel adds the 64 partial hashes and
// returns at once when the sum equals the hash stored with the table; otherwise all workgroups rebuild their entries and the last one
// to finish publishes the new hash.  No host-side version bookkeeping can go stale (in-place writes through .data, module.to(),
// load_state_dict, optimizer steps all change the content), no host synchronisation, capturable in a graph.  Cost when the table is
// current: two launches of a few microseconds (a single-kernel form in which every workgroup hashed all 260 KB took 80 us).
constexpr int kHashParts = 64;
struct TableState { unsigned long long fp; unsigned int done; unsigned int pad; unsigned long long part[kHashParts]; };
constexpr int kTabE = 8;              // table entries per workgroup

__device__ __forceinline__ unsigned long long mix64(unsigned long long h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
  return h;
}

__global__ void __launch_bounds__(256) embedding_table_hash_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                                   const float* __restrict__ div_term, int C, TableState* __restrict__ st) {
  __shared__ unsigned long long part[4];
  unsigned long long h = 0;
  const int nW = C * C, total = nW + C + C / 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += kHashParts * 256) {
    const float v = i < nW ? W[i] : (i < nW + C ? b[i - nW] : div_term[i - nW - C]);
    h += mix64((unsigned long long)__float_as_uint(v) | ((unsigned long long)(unsigned)(i + 1) << 32));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)h, o), hi = __shfl_xor((unsigned)(h >> 32), o);
    h += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) st->part[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(256) embedding_table_refresh_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                                      const float* __restrict__ div_term, int C, int entries,
                                                                      float per_unit, float2* __restrict__ tab,
                                                                      TableState* __restrict__ st) {
  extern __shared__ double lds_d[];                       // emb[kTabE][C], demb[kTabE][C]
  __shared__ unsigned long long fp_s;
  // 1. content hash = sum of the partial hashes (embedding_table_hash_kernel, same stream)
  if (threadIdx.x < 64) {
    unsigned long long h = st->part[threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)h, o), hi = __shfl_xor((unsigned)(h >> 32), o);
      h += ((unsigned long long)hi << 32) | lo;
    }
    if (threadIdx.x == 0)
      fp_s = mix64(h ^ ((unsigned long long)(unsigned)entries << 32) ^ (unsigned long long)__float_as_uint(per_unit) ^
                   ((unsigned long long)(unsigned)C << 20)) | 1ULL;      // never 0: a zero-initialised state never matches
  }
  __syncthreads();
  const unsigned long long fp = fp_s;
  if (__hip_atomic_load(&st->fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fp) return;
  // 2. rebuild this workgroup's entries
  double* emb = lds_d;
  double* demb = lds_d + kTabE * C;
  const int e0 = blockIdx.x * kTabE;
  for (int t = threadIdx.x; t < kTabE * (C / 2); t += 256) {
    const int e = t / (C / 2), i = t - e * (C / 2);
    const double w = (double)div_term[i];
    const double om = ((double)(e0 + e) / (double)per_unit) * w;
    double s, c;
    sincos(om, &s, &c);
    emb[e * C + 2 * i] = s;      emb[e * C + 2 * i + 1] = c;
    demb[e * C + 2 * i] = c * w; demb[e * C + 2 * i + 1] = -s * w;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += 256) {
    double f[kTabE], df[kTabE];
#pragma unroll
    for (int e = 0; e < kTabE; e++) { f[e] = 0.0; df[e] = 0.0; }
    const float* row = W + (size_t)o * C;
    for (int c = 0; c < C; c += 2) {
      const float2 w2 = *reinterpret_cast<const float2*>(row + c);
      const double w0 = w2.x, w1 = w2.y;
#pragma unroll
      for (int e = 0; e < kTabE; e++) {
        f[e] += w0 * emb[e * C + c] + w1 * emb[e * C + c + 1];
        df[e] += w0 * demb[e * C + c] + w1 * demb[e * C + c + 1];
      }
    }
    const double bo = (double)b[o];
#pragma unroll
    for (int e = 0; e < kTabE; e++)
      if (e0 + e < entries) tab[(size_t)(e0 + e) * C + o] = make_float2((float)(f[e] + bo), (float)df[e]);
  }
  // 3. the last workgroup to finish publishes the hash.  ONE writer per (table, state): the callers keep a table per launch stream
  //    (se3et_amd/ops.py _embedding_table) -- two streams rebuilding one table at once would mix their completion counts
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(&st->done, 1u);
    if (prev % gridDim.x == gridDim.x - 1) __hip_atomic_store(&st->fp, fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

extern "C" size_t se3_embedding_table_state_bytes(void) { return sizeof(TableState); }

extern "C" int se3_embedding_table_refresh(const float* weight, const float* bias, const float* div_term, int C, int entries,
                                           float entries_per_unit, float* table, void* state, void* stream) {
  SE3_REQUIRE(weight && bias && div_term && table && state, SE3_ERR_INVALID_ARG, "embedding_table_refresh: null pointer");
  SE3_REQUIRE(C >= 2 && C % 2 == 0 && C <= 1024 && entries >= 2 && entries_per_unit > 0.f, SE3_ERR_INVALID_ARG,
              "embedding_table_refresh: bad sizes");
  const unsigned grid = (unsigned)((entries + kTabE - 1) / kTabE);
  const size_t lds = (size_t)2 * kTabE * C * sizeof(double);
  embedding_table_hash_kernel<<<kHashParts, 256, 0, (hipStream_t)stream>>>(weight, bias, div_term, C, static_cast<TableState*>(state));
  embedding_table_refresh_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(weight, bias, div_term, C, entries, entries_per_unit,
                                                                         reinterpret_cast<float2*>(table),
                                                                         static_cast<TableState*>(state));
  SE3_CHECK_LAUNCH("embedding_table_refresh");
  return SE3_OK;
}

// record workspace of the channel-slice form: (N N) int4 intervals + (N N, 4) float4 Hermite weights
extern "C" size_t se3_geo_embedding_workspace_bytes(int N) { return (size_t)N * N * (sizeof(int4) + 4 * sizeof(float4)); }

static int geo_embedding(const float* points, const int64_t* knn, int N, int C, const float* table_d, int d_entries,
                         float d_entries_per_unit, const float* table_a, int a_entries, float a_entries_per_unit,
                         float sigma_d, float sigma_a, const float* w_d, const float* b_d, const float* w_a, const float* b_a,
                         const float* div_term, const float* wigner_d1, int num_anchors, void* emb, int emb_bf16,
                         float* eq_emb, void* workspace, size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(points && knn && table_d && table_a && w_d && b_d && w_a && b_a && div_term && emb, SE3_ERR_INVALID_ARG,
              "geo_embedding: null pointer");
  SE3_REQUIRE(N >= 1 && C >= 2 && C % 2 == 0 && d_entries >= 2 && a_entries >= 2, SE3_ERR_INVALID_ARG, "geo_embedding: bad sizes");
  SE3_REQUIRE((eq_emb == nullptr) || (wigner_d1 != nullptr && num_anchors >= 1), SE3_ERR_INVALID_ARG,
              "geo_embedding: equivariant output needs the Wigner table");
  EmbParams P;
  P.sigma_d_inv = 1.0f / sigma_d;
  P.factor_a = 180.0f / (sigma_a * 3.14159265358979323846f);
  P.d_inv_h = d_entries_per_unit; P.a_inv_h = a_entries_per_unit;
  P.d_entries = d_entries; P.a_entries = a_entries;
  // channel-slice form when the angle table slice fits in LDS (SE3ET: 418 entries x 256 B) and the caller brought the record workspace
  // (workspace = NULL selects the single-kernel form above)
  const size_t slice_lds = (size_t)a_entries * 16 * sizeof(float4) + (size_t)(kSliceThreads / 64) * 2 * kRecBytes;
  if (C % kCS == 0 && slice_lds <= 150 * 1024 && workspace != nullptr) {
    SE3_REQUIRE(workspace_bytes >= se3_geo_embedding_workspace_bytes(N), SE3_ERR_WORKSPACE, "geo_embedding: workspace too small");
    SE3_REQUIRE((long long)N * N < (1ll << 31) / 4, SE3_ERR_UNSUPPORTED, "geo_embedding: N = %d too large for the record index", N);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(geo_embedding_slice_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(geo_embedding_slice_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr_set = true;
    }
    const size_t pairs = (size_t)N * N;
    int4* jrec = static_cast<int4*>(workspace);
    float4* wrec = reinterpret_cast<float4*>(jrec + pairs);
    se3geo::launch_pair_terms(points, knn, N, P, wigner_d1, jrec, wrec, eq_emb, num_anchors, (hipStream_t)stream);
    // one workgroup per CU (LDS): row blocks so that slices x blocks is a multiple of the 256 CUs when N allows
    const int slices = C / kCS;
    int row_blocks = 256 / slices > 0 ? 256 / slices : 1;
    if (row_blocks > N) row_blocks = N;
    const int rows_per_block = (N + row_blocks - 1) / row_blocks;
    row_blocks = (N + rows_per_block - 1) / rows_per_block;
    const ExactArgs X{w_d, b_d, w_a, b_a, div_term};
    const dim3 grid((unsigned)slices, (unsigned)row_blocks);
    if (emb_bf16)
      geo_embedding_slice_kernel<true><<<grid, kSliceThreads, slice_lds, (hipStream_t)stream>>>(
          N, C, reinterpret_cast<const float2*>(table_d), reinterpret_cast<const float2*>(table_a), a_entries, jrec, wrec, X, emb, rows_per_block);
    else
      geo_embedding_slice_kernel<false><<<grid, kSliceThreads, slice_lds, (hipStream_t)stream>>>(
          N, C, reinterpret_cast<const float2*>(table_d), reinterpret_cast<const float2*>(table_a), a_entries, jrec, wrec, X, emb, rows_per_block);
    SE3_CHECK_LAUNCH("geo_embedding_slice");
    return SE3_OK;
  }
  int split = (1024 + N - 1) / N;
  if (split < 1) split = 1;
  if (split > (N + kMB - 1) / kMB) split = (N + kMB - 1) / kMB;
  dim3 grid((unsigned)N, (unsigned)split);
  const int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
  geo_embedding_kernel<<<grid, threads, 0, (hipStream_t)stream>>>(
      points, knn, N, C, reinterpret_cast<const float2*>(table_d), reinterpret_cast<const float2*>(table_a), P, w_d, b_d, w_a,
      b_a, div_term, wigner_d1, emb, emb_bf16, eq_emb, num_anchors);
  SE3_CHECK_LAUNCH("geo_embedding");
  return SE3_OK;
}

extern "C" int se3_geo_embedding_fwd(const float* points, const int64_t* knn, int N, int C, const float* table_d,
                                     int d_entries, float d_entries_per_unit, const float* table_a, int a_entries,
                                     float a_entries_per_unit, float sigma_d, float sigma_a, const float* w_d,
                                     const float* b_d, const float* w_a, const float* b_a, const float* div_term,
                                     const float* wigner_d1, int num_anchors, float* emb, float* eq_emb, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return geo_embedding(points, knn, N, C, table_d, d_entries, d_entries_per_unit, table_a, a_entries, a_entries_per_unit, sigma_d,
                       sigma_a, w_d, b_d, w_a, b_a, div_term, wigner_d1, num_anchors, emb, 0, eq_emb, workspace, workspace_bytes, stream);
}

extern "C" int se3_geo_embedding_bf16_fwd(const float* points, const int64_t* knn, int N, int C, const float* table_d,
                                          int d_entries, float d_entries_per_unit, const float* table_a, int a_entries,
                                          float a_entries_per_unit, float sigma_d, float sigma_a, const float* w_d,
                                          const float* b_d, const float* w_a, const float* b_a, const float* div_term,
                                          const float* wigner_d1, int num_anchors, uint16_t* emb, float* eq_emb,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  return geo_embedding(points, knn, N, C, table_d, d_entries, d_entries_per_unit, table_a, a_entries, a_entries_per_unit, sigma_d,
                       sigma_a, w_d, b_d, w_a, b_a, div_term, wigner_d1, num_anchors, emb, 1, eq_emb, workspace, workspace_bytes, stream);
}

// Operands of the weight gradients of the embedding (training step): S (4, N, N, C) = emb(index_t) for t = distance, angle 0..2 and dEk
// (3, N, N, C) = grad_emb masked to the channels where angle k is the arg-max.  Then dW_d = grad_emb^T S[0], dW_a = sum_k dEk[k]^T S[1 + k]
// (library GEMMs), db_d = db_a = sum grad_emb.
extern "C" int se3_geo_embedding_bwd_operands(const float* points, const int64_t* knn, int N, int C, const float* table_a, int a_entries,
                                              float a_entries_per_unit, float sigma_d, float sigma_a, const float* w_a, const float* b_a,
                                              const float* div_term, const float* grad_emb, float* S, float* dEk, void* stream) {
  SE3_REQUIRE(points && knn && table_a && w_a && b_a && div_term && grad_emb && S && dEk, SE3_ERR_INVALID_ARG,
              "geo_embedding_bwd_operands: null pointer");
  SE3_REQUIRE(N >= 1 && C >= 2 && C % 2 == 0 && a_entries >= 2, SE3_ERR_INVALID_ARG, "geo_embedding_bwd_operands: bad sizes");
  EmbParams P;
  P.sigma_d_inv = 1.0f / sigma_d;
  P.factor_a = 180.0f / (sigma_a * 3.14159265358979323846f);
  P.d_inv_h = 1.f; P.a_inv_h = a_entries_per_unit;
  P.d_entries = 2; P.a_entries = a_entries;
  int split = (1024 + N - 1) / N;
  if (split < 1) split = 1;
  if (split > (N + kMB - 1) / kMB) split = (N + kMB - 1) / kMB;
  const int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
  geo_embedding_bwd_operands_kernel<<<dim3((unsigned)N, (unsigned)split), threads, 0, (hipStream_t)stream>>>(
      points, knn, N, C, reinterpret_cast<const float2*>(table_a), P, w_a, b_a, div_term, grad_emb, S, dEk);
  SE3_CHECK_LAUNCH("geo_embedding_bwd_operands");
  return SE3_OK;
}
