// Micro-benchmark (not part of the library): sustained rate of v_mfma_f32_16x16x32_bf16 in the bf16x6 pattern of
// csrc/kpconv_contract.hip (six accumulators, six products of three operand pieces), random operands, 1 / 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16_rate.hip -o /tmp/mfma_bf16_rate && /tmp/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const uint4* __restrict__ in, float* out, int iters) {
  f32x4 acc[6];
  for (int c = 0; c < 6; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[3][6], b[3];
  for (int p = 0; p < 3; p++) {
    for (int r = 0; r < 6; r++) a[p][r] = __builtin_bit_cast(bf16x8, in[(p * 6 + r) * 64 + (threadIdx.x & 63)]);
    b[p] = __builtin_bit_cast(bf16x8, in[(18 + p) * 64 + (threadIdx.x & 63)]);
  }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][r], b[0], acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][r], b[2], acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][r], b[1], acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][r], b[0], acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][r], b[1], acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][r], b[0], acc[r], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < 6; c++) for (int r = 0; r < 4; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  uint4* in; float* out;
  hipMalloc(&in, 21 * 64 * sizeof(uint4)); hipMalloc(&out, 256 * 1024 * sizeof(float));
  unsigned short h[21 * 64 * 8];
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));     // random bf16 around +-1
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int wpc : {4, 8}) {
    for (int zero = 0; zero < 2; zero++) {
      if (zero) hipMemset(in, 0, 21 * 64 * sizeof(uint4));
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      k<<<256, 64 * wpc>>>(in, out, iters); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 3; i++) k<<<256, 64 * wpc>>>(in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      const double n = (double)iters * 36;
      printf("%s operands, %d wave(s)/SIMD: %.3f ms, %.1f ns per MFMA per SIMD, %.2f PFLOP/s bf16 = %.0f TFLOP/s f32-equivalent (x6)\n",
             zero ? "zero  " : "random", wpc / 4, ms, ms * 1e6 / n / (wpc / 4.0), n * 16384 * wpc * 256 / (ms * 1e-3) / 1e15,
             n * 16384 * wpc * 256 / (ms * 1e-3) / 1e12 / 6);
    }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  }
  return 0;
}
