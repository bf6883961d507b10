// Diagnostic build of the fused KPConv kernel with wave time stamps (s_memtime at the phase boundaries of the first 64 workgroups).
// Not part of libse3et_hip.so: builds its own library from the product source with the SE3_STAMP hook defined below.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/micro/kpconv_stamps.hip se3et_amd/csrc/capi_common.hip -o tools/micro/libkpconv_stamps.so
// -DSE3_DIAG_FIXED_B: every K16-step multiplies with the first step's weight fragments (wrong results; shows what the weight stream costs)
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ long long* g_stamps = nullptr;
constexpr int kStampBlocks = 64, kStampSteps = 40, kStampSlots = 8;
#define SE3_STAMP(step_, slot_)                                                                                            \
  if (g_stamps && blockIdx.x < kStampBlocks && blockIdx.y == 0 && (step_) < kStampSteps && lane == 0)                      \
    g_stamps[(((int64_t)blockIdx.x * 16 + wave) * kStampSteps + (step_)) * kStampSlots + (slot_)] = __builtin_amdgcn_s_memtime();
#ifdef SE3_DIAG_FIXED_B
#define SE3_DIAG_WEIGHT_STEP(gs_, ksp_) (ksp_)
#endif
#include "../../se3et_amd/csrc/kpconv_mfma.hip"

extern "C" int se3_debug_kpconv_set_stamps(void* device_buffer) {      // 64 blocks x 16 waves x 40 steps x 8 slots x int64
  long long* p = static_cast<long long*>(device_buffer);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
