// Micro-benchmark (not part of the library): issue rate of f32 MFMAs on gfx950, dependent vs independent chains.
// hipcc --offload-arch=gfx950 -O3 tests/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void k32(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void k16(float* out, int iters, float a, float b) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 4; r++) acc[c][r] = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 4; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_it(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; i++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 16 * sizeof(float));
  const int iters = 4000;     // x 8 x CHAINS MFMAs per wave
  for (int wpc : {4, 8, 16}) {           // waves per CU (1, 2, 4 per SIMD) on 256 CUs
    dim3 grid(256), block(64 * wpc);
    auto report = [&](const char* name, float ms, int chains, double flop_per) {
      const double n = (double)iters * 8 * chains;                // MFMAs per wave
      const double cyc = ms * 1e-3 * 2.4e9 / n / (wpc / 4.0);     // cycles per MFMA per SIMD at 2.4 GHz
      const double tf = n * flop_per * wpc * 256 / (ms * 1e-3) / 1e12;
      printf("%-28s waves/SIMD %d chains %d: %.3f ms  %.1f cycles/MFMA/SIMD (at 2.4 GHz)  %.1f TFLOP/s\n", name, wpc / 4, chains, ms, cyc, tf);
    };
    report("mfma_f32_32x32x2", time_it([&] { k32<1><<<grid, block>>>(out, iters, 1.f, 2.f); }), 1, 4096);
    report("mfma_f32_32x32x2", time_it([&] { k32<2><<<grid, block>>>(out, iters, 1.f, 2.f); }), 2, 4096);
    report("mfma_f32_16x16x4", time_it([&] { k16<1><<<grid, block>>>(out, iters, 1.f, 2.f); }), 1, 2048);
    report("mfma_f32_16x16x4", time_it([&] { k16<2><<<grid, block>>>(out, iters, 1.f, 2.f); }), 2, 2048);
    report("mfma_f32_16x16x4", time_it([&] { k16<4><<<grid, block>>>(out, iters, 1.f, 2.f); }), 4, 2048);
  }
  return 0;
}
