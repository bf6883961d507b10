// The record kernel of the geometric embedding's channel-slice form (csrc/geo_embedding.hip: geo_pair_terms_kernel), in a translation unit
// of its own that is compiled WITHOUT the SLP vectoriser (build.py: SE3_NO_SLP_VECTORIZE below).  The vectoriser packs independent scalar
// f32 chains into v_pk_*_f32 instructions with op_sel shuffles, and one such sequence -- a packed instruction whose LOW result reads the HIGH
// half of the packed product written two instructions earlier -- returned wrong values on MI355X (geo_records.h: hermite_weights; DESIGN.md
// section 7).  That sequence is written out by hand there; this kernel's cross products produced six more shuffles of the same shape
// (v_pk_mov_b32 ... op_sel:[1,0] on a fresh packed result, never seen failing) -- the kernel runs 20 us per cloud and is bound by its stores,
// so it simply does without packed arithmetic.  tools/scan_pk_f32_forwarding.py checks the ISA of the whole library for the shape.
// SE3_NO_SLP_VECTORIZE
#include "geo_records.h"

namespace se3geo {
namespace {

__device__ __forceinline__ void pair_term_record(float x, float inv_h, int entries, int& j_out, float4& w_out) {
  const float u = x * inv_h;
  const int j = (int)floorf(u);
  const bool ok = (j >= 0) && (j + 1 < entries);
  const float tt = u - (float)j, h = 1.0f / inv_h;
  j_out = ok ? j : -1;
  // outside the table: the index itself, for the exact evaluation
  w_out = ok ? hermite_weights(tt, h) : make_float4(x, 0.f, 0.f, 0.f);
}

// one thread per (n, m): the four indices, their table intervals and Hermite weights (80 contiguous bytes of records), and Eeq
__global__ __launch_bounds__(256) void geo_pair_terms_kernel(const float* __restrict__ pts, const int64_t* __restrict__ knn, int N,
                                                             EmbParams P, const float* __restrict__ wigner_d1, int4* __restrict__ jrec,
                                                             float4* __restrict__ wrec, float* __restrict__ eq_emb, int A) {
  const unsigned pair = blockIdx.x * 256u + threadIdx.x;
  if (pair >= (unsigned)N * (unsigned)N) return;
  const int n = (int)(pair / (unsigned)N), m = (int)(pair - (unsigned)n * (unsigned)N);
  const float px = pts[3 * n], py = pts[3 * n + 1], pz = pts[3 * n + 2];
  const float qx = pts[3 * m], qy = pts[3 * m + 1], qz = pts[3 * m + 2];
  const float vx = qx - px, vy = qy - py, vz = qz - pz;
  int j[4];
  float4 w[4];
  {
    const float nn2 = se3_ref_sq_norm(px, py, pz);
    const float d2 = se3_ref_sq_dist(px, py, pz, nn2, qx, qy, qz, se3_ref_sq_norm(qx, qy, qz));
    pair_term_record(sqrtf(d2) * P.sigma_d_inv, P.d_inv_h, P.d_entries, j[0], w[0]);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int64_t jn = knn[3 * n + k];
    const float rx = pts[3 * jn] - px, ry = pts[3 * jn + 1] - py, rz = pts[3 * jn + 2] - pz;
    const float cx = ry * vz - rz * vy, cy = rz * vx - rx * vz, cz = rx * vy - ry * vx;
    const float sn = sqrtf(cx * cx + cy * cy + cz * cz);
    float cs = rx * vx + ry * vy + rz * vz;
    cs = (cs == 0.f) ? 0.f : cs;        // see geo_embedding_kernel
    pair_term_record(atan2f(sn, cs) * P.factor_a, P.a_inv_h, P.a_entries, j[1 + k], w[1 + k]);
  }
  jrec[pair] = make_int4(j[0], j[1], j[2], j[3]);
#pragma unroll
  for (int t = 0; t < 4; t++) wrec[(size_t)pair * 4 + t] = w[t];
  if (eq_emb != nullptr) {
    // unit vector of p_n - p_m (zero vector -> 0, as F.normalize with eps 1e-12)
    const float len = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.f / fmaxf(len, 1e-12f);
    const float ux = -vx * inv, uy = -vy * inv, uz = -vz * inv;
    const float c1 = 0.4886025119029199f;        // sqrt(3 / (4 pi))
    for (int a = 0; a < A; a++) {
      const float* D = wigner_d1 + 9 * a;        // D^1_a (3, 3): out_c = sum_d D[c][d] Y1_d
      float4 o;
      o.x = 0.28209479177387814f;                // 1 / (2 sqrt(pi))
      o.y = c1 * (D[0] * ux + D[1] * uy + D[2] * uz);
      o.z = c1 * (D[3] * ux + D[4] * uy + D[5] * uz);
      o.w = c1 * (D[6] * ux + D[7] * uy + D[8] * uz);
      reinterpret_cast<float4*>(eq_emb)[(size_t)a * N * N + pair] = o;
    }
  }
}

}  // namespace

void launch_pair_terms(const float* pts, const int64_t* knn, int N, EmbParams P, const float* wigner_d1, int4* jrec, float4* wrec, float* eq_emb,
                       int A, hipStream_t stream) {
  const size_t pairs = (size_t)N * N;
  geo_pair_terms_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, stream>>>(pts, knn, N, P, wigner_d1, jrec, wrec, eq_emb, A);
}

}  // namespace se3geo
