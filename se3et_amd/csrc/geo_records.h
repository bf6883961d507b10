// Shared by geo_embedding.hip and geo_records.hip: the index parameters of the geometric embedding and the Hermite weights of a table interval.
#pragma once
#include "common.h"

namespace se3geo {

struct EmbParams {
  float sigma_d_inv, factor_a;        // index scales
  float d_inv_h, a_inv_h;             // table resolutions (entries per index unit)
  int d_entries, a_entries;           // table lengths
};

// (h00, h01, h h10, h h11)(t): the four cubic Hermite weights of a table interval, as SCALAR instruction chains kept apart by empty asm
// statements.  Left to the vectoriser, (2 t^3, 3 t^2) became one v_pk_mul_f32 and the next-but-one instruction a v_pk_fma_f32 whose LOW
// result reads the HIGH half of that product through op_sel:
//     v_pk_mul_f32 v[0:1], v[14:15], s[20:21]
//     v_sub_f32    v21, v14, v15
//     v_pk_fma_f32 v[18:19], v[14:15], s[20:21], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]
// On the MI355X boxes of the pool that read returned 0 for lanes 48..63 of a wave -- h00 = 2 t^3 + 1 instead of 2 t^3 - 3 t^2 + 1, every
// other lane and component right -- about once per 10^8 records, and only in the first forwards after the GPU had idled with several
// streams starting at once: the intermittent difference of tests/test_gpu_concurrency.py (tools/concurrency_bisect.py records the
// evidence).  No other kernel of the library contains the sequence (tools/scan_pk_f32_forwarding.py).
__device__ __forceinline__ float4 hermite_weights(float tt, float h) {
  const float t2 = tt * tt, t3 = t2 * tt;
  float a = 2.f * t3, b = 3.f * t2;
  asm volatile("" : "+v"(a));
  asm volatile("" : "+v"(b));
  float wx = (a - b) + 1.f;
  asm volatile("" : "+v"(wx), "+v"(a), "+v"(b));
  float wy = __builtin_fmaf(3.f, t2, -a);
  asm volatile("" : "+v"(wy));
  float wz = h * ((t3 - 2.f * t2) + tt);
  asm volatile("" : "+v"(wz));
  return make_float4(wx, wy, wz, h * (t3 - t2));
}

// geo_records.hip: the (N N) record arrays of the channel-slice form (+ the equivariant embedding); launched by geo_embedding.hip
void launch_pair_terms(const float* pts, const int64_t* knn, int N, EmbParams P, const float* wigner_d1, int4* jrec, float4* wrec, float* eq_emb,
                       int A, hipStream_t stream);

}  // namespace se3geo
