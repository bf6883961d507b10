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

namespace {

struct EmbParams {
  float sigma_d_inv, factor_a;        // index scales
  float d_inv_h, a_inv_h;             // table resolutions (entries per index unit)
  int d_entries, a_entries;           // table lengths
};

__device__ __forceinline__ float hermite(const float2* __restrict__ tab, int C, int c, float x, float inv_h, int entries,
                                         bool& ok) {
  const float u = x * inv_h;
  int j = (int)floorf(u);
  ok = (j >= 0) && (j + 1 < entries);
  j = min(max(j, 0), entries - 2);
  const float t = u - (float)j;
  const float2 p0 = tab[(size_t)j * C + c], p1 = tab[(size_t)(j + 1) * C + c];
  const float h = 1.0f / inv_h;
  const float t2 = t * t, t3 = t2 * t;
  const float h00 = 2.f * t3 - 3.f * t2 + 1.f, h10 = t3 - 2.f * t2 + t, h01 = -2.f * t3 + 3.f * t2, h11 = t3 - t2;
  return (h00 * p0.x + h01 * p1.x) + h * (h10 * p0.y + h11 * p1.y);
}

// exact evaluation W[c, :] . emb(x) + b[c] (fallback for indices outside the table)
__device__ float exact_eval(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ div_term,
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
  const float nn2 = px * px + py * py + pz * pz;
  for (int m0 = m_begin; m0 < m_end; m0 += kMB) {
    __syncthreads();
    if (threadIdx.x < kMB) {
      const int m = min(m0 + (int)threadIdx.x, N - 1);
      const float qx = pts[3 * m], qy = pts[3 * m + 1], qz = pts[3 * m + 2];
      const float d2 = fmaxf(nn2 - 2.f * (px * qx + py * qy + pz * qz) + (qx * qx + qy * qy + qz * qz), 0.f);
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
        const float t2 = tt * tt, t3 = t2 * tt;
        idx_s[threadIdx.x][t] = x[t];
        j_s[threadIdx.x][t] = ok ? j : -1;
        wt_s[threadIdx.x][t] = make_float4(2.f * t3 - 3.f * t2 + 1.f, -2.f * t3 + 3.f * t2, h * (t3 - 2.f * t2 + tt), h * (t3 - t2));
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
  // 3. the last workgroup to finish publishes the hash (modulo: concurrent rebuilds of the same table stay consistent)
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

static int geo_embedding(const float* points, const int64_t* knn, int N, int C, const float* table_d, int d_entries,
                         float d_entries_per_unit, const float* table_a, int a_entries, float a_entries_per_unit,
                         float sigma_d, float sigma_a, const float* w_d, const float* b_d, const float* w_a, const float* b_a,
                         const float* div_term, const float* wigner_d1, int num_anchors, void* emb, int emb_bf16,
                         float* eq_emb, void* stream) {
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
                                     const float* wigner_d1, int num_anchors, float* emb, float* eq_emb, void* stream) {
  return geo_embedding(points, knn, N, C, table_d, d_entries, d_entries_per_unit, table_a, a_entries, a_entries_per_unit, sigma_d,
                       sigma_a, w_d, b_d, w_a, b_a, div_term, wigner_d1, num_anchors, emb, 0, eq_emb, stream);
}

extern "C" int se3_geo_embedding_bf16_fwd(const float* points, const int64_t* knn, int N, int C, const float* table_d,
                                          int d_entries, float d_entries_per_unit, const float* table_a, int a_entries,
                                          float a_entries_per_unit, float sigma_d, float sigma_a, const float* w_d,
                                          const float* b_d, const float* w_a, const float* b_a, const float* div_term,
                                          const float* wigner_d1, int num_anchors, uint16_t* emb, float* eq_emb,
                                          void* stream) {
  return geo_embedding(points, knn, N, C, table_d, d_entries, d_entries_per_unit, table_a, a_entries, a_entries_per_unit, sigma_d,
                       sigma_a, w_d, b_d, w_a, b_a, div_term, wigner_d1, num_anchors, emb, 1, eq_emb, stream);
}
