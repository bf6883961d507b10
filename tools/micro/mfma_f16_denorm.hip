// Does v_mfma_f32_32x32x16_f16 honour subnormal f16 inputs, and does v_cvt_f16_f32 produce them?  (The lo pieces of the KPConv / attention
// operand splits are subnormal for small values.)   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f16_denorm.hip -o tools/micro/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void k(float a_val, float b_val, float* out) {
  f16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8); float h[2];
  const float vals[] = {1.0f, 6.2e-5f, 3.0e-5f, 1.0e-6f, 6.0e-8f};
  for (float v : vals) {
    k<<<1, 64>>>(v, 1.0f, d); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.3e: cvt -> %.6e, mfma (K = 16, b = 1) -> %.6e (expected %.6e)\n", v, h[1], h[0], 16.0 * h[1]);
    k<<<1, 64>>>(1.0f, v, d); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("b = %.3e:                      mfma -> %.6e\n", v, h[0]);
  }
  return 0;
}
