// Diagnostic build of the fused KPConv kernel with wave time stamps (s_memtime at the phase boundaries of the first 64 workgroups).
// Not part of libse3et_hip.so: builds its own library from the product source with SE3_KPCONV_STAMPS defined.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/micro/kpconv_stamps.hip se3et_amd/csrc/capi_common.hip -o tools/micro/libkpconv_stamps.so
#define SE3_KPCONV_STAMPS 1
// -DSE3_DIAG_FIXED_B: every K16-step multiplies with the first step's weight fragments (wrong results; shows what the weight stream costs)
#include "../../se3et_amd/csrc/kpconv_mfma.hip"

extern "C" int se3_debug_kpconv_set_stamps(void* device_buffer) {      // 64 blocks x 16 waves x 40 steps x 8 slots x int64
  long long* p = static_cast<long long*>(device_buffer);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
