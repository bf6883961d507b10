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
