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
