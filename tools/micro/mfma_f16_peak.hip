// Micro-benchmark (not part of the library): the f16 MFMA rate the chip SUSTAINS with every SIMD issuing back-to-back independent MFMAs
// (v_mfma_f32_32x32x16_f16, v_mfma_f32_16x16x32_f16) -- the ceiling the KPConv / dense kernels are priced against at the real clock.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f16_peak.hip -o tools/micro/mfma_f16_peak && tools/micro/mfma_f16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void k32(float* out, int iters, _Float16 a0, _Float16 b0, long long* clk) {
  f32x16 acc[CHAINS];
  f16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = a0; b[i] = b0; }
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int CHAINS>
__global__ void k16(float* out, int iters, _Float16 a0, _Float16 b0, long long* clk) {
  f32x4 acc[CHAINS];
  f16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = a0; b[i] = b0; }
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 4; r++) acc[c][r] = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 4; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 16 * sizeof(float));
  long long* clk; hipMalloc(&clk, 16);
  const int iters = 20000;
  for (int wpc : {4, 8}) {
    dim3 grid(256), block(64 * wpc);
    for (int which = 0; which < 2; which++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&] {
        if (which == 0) k32<4><<<grid, block>>>(out, iters, (_Float16)1.f, (_Float16)0.5f, clk);
        else k16<4><<<grid, block>>>(out, iters, (_Float16)1.f, (_Float16)0.5f, clk);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; r++) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double n = (double)iters * 4 * 4;                              // MFMAs per wave
      const double flop = which == 0 ? 32768.0 : 16384.0;
      const double tf = n * flop * wpc * 256 / (ms * 1e-3) / 1e12;
      long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      printf("%-26s waves/SIMD %d: %.3f ms  %.0f TFLOP/s = %.2f of 2500", which == 0 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", wpc / 4, ms, tf, tf / 2500);
      if (which == 0) printf("   shader clock during the kernel: %.0f MHz (clock64 / wall_clock64 at 100 MHz), %.1f shader cycles per MFMA and SIMD", (double)h[0] / h[1] * 100.0, (double)h[0] / n / (wpc / 4.0));
      printf("\n");
    }
  }
  return 0;
}
