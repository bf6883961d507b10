// Shared helpers for the gfx950 kernels of libse3et_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/se3et_hip.h"

#define SE3_WAVE 64

void se3_set_error(const char* fmt, ...);

#define SE3_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      se3_set_error(__VA_ARGS__);           \
      return (code);                        \
    }                                       \
  } while (0)

#define SE3_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      se3_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return SE3_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

static inline int64_t se3_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Exactly-rounded float32 primitives for the bit-exact geometry kernels.  hipcc's __f*_rn intrinsics are plain
// operators defined in a compiler header (and __fsqrt_rn is the approximate native sqrt), so: sources that define
// SE3_EXACT_FP are compiled with -ffp-contract=off (se3et_amd/build.py) so that products and sums stay unfused, and
// division and square root go through float64 (53 >= 2*24+2 bits makes the double result round to the correctly
// rounded float32 result).
__device__ __forceinline__ float se3_exact_div(float a, float b) { return (float)((double)a / (double)b); }
__device__ __forceinline__ float se3_exact_sqrt(float a) { return (float)sqrt((double)a); }

// Squared distance of two 3-D points as geotransformer/modules/ops/pairwise_distance.py:4-30 evaluates it in float32 -- x2 - 2 xy + y2 with
// x2 = (x^2 + y^2) + z^2 (torch.sum over the last dim, unfused), xy the K = 3 matmul = fma(z, z', fma(y, y', x x')) (k ascending, fused:
// bit-identical to torch.matmul on the build container's CPU for all 168 100 pairs of the demo pair's superpoints), (x2 - 2 xy) + y2,
// clamped at 0.  The expression cancels: at coordinates of ~3 m its rounding noise is ~2e-6 in d^2, i.e. a SELF distance of up to
// 1.4 mm instead of 0 and +-5e-3 on the distance index of the geometric embedding (d / sigma_d, sigma_d = 0.2) -- the reference's
// embedding of near-coincident superpoints is a function of this noise, so it is restated operation by operation, not approximated (found on data/demo: 2.6e-2 on the first transformer layer with the products fused differently).
// (hipcc's __fmul_rn / __fadd_rn are plain operators -- the contraction pass fuses them like any a * b + c --, so the unfused steps
// are fenced with the scoped pragma, which tags the instructions themselves and survives inlining.)
__device__ __forceinline__ float se3_ref_sq_norm(float x, float y, float z) {
#pragma clang fp contract(off)
  float xx = x * x, yy = y * y;
  asm volatile("" : "+v"(xx), "+v"(yy));          // (scalar products: the SLP vectoriser would pack x x and y y into one v_pk_mul_f32 followed by
  const float zz = z * z;                          // an op_sel shuffle of its fresh result -- the shape tests/test_isa_hazard.py keeps out of the library)
  const float s = xx + yy;
  return s + zz;
}
__device__ __forceinline__ float se3_ref_sq_dist(float px, float py, float pz, float p2, float qx, float qy, float qz, float q2) {
#pragma clang fp contract(off)
  const float xx = px * qx;
  const float dot = __builtin_fmaf(pz, qz, __builtin_fmaf(py, qy, xx));
  const float two = 2.f * dot;
  const float a = p2 - two;
  const float b = a + q2;
  return fmaxf(b, 0.f);
}

// Order-independent accumulation (the training step's scatter-adds): contributions are added as 64-bit FIXED-POINT integers -- integer
// addition is associative, so the sum does not depend on the arrival order of the atomics and two runs are bit-identical.  The scale comes
// from an upper bound of the contributions (*bound: max |gradient| of the call, a DEVICE word), `growth` = log2 of how much one contribution
// can exceed it, and the number of contributions that can meet in one element (<= terms): 2^(61 - e) with 2^e >= 2^growth terms bound cannot
// overflow and resolves bound 2^-(54 - growth - log2 terms) -- below the float32 rounding of the sum the float atomics produce.
__device__ __forceinline__ int se3_fixed_scale_exp(const float* bound, int64_t terms, int growth) {
  const unsigned b = __float_as_uint(*bound);
  const int eb = (int)((b >> 23) & 0xff) - 126;                          // bound < 2^eb
  int lg = 1;
  while ((1ll << lg) < terms + 1) lg++;
  return 61 - (eb + growth + lg);
}
__device__ __forceinline__ void se3_fixed_add(unsigned long long* acc, float v, double scale) {
  const long long q = __double2ll_rn((double)v * scale);
  if (q != 0) atomicAdd(acc, (unsigned long long)q);                      // (two's complement: unsigned wrap-around addition is signed addition)
}

__device__ __forceinline__ int se3_lane() { return threadIdx.x & (SE3_WAVE - 1); }

__device__ __forceinline__ float se3_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float se3_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
