// Micro-benchmark (not part of the library): the multiply half of csrc/kpconv_contract.hip in isolation -- per step 18 ds_read_b128
// fragment reads + 36 v_mfma_f32_16x16x32_bf16, 8 waves per workgroup, one workgroup per CU; variants: barrier per step or not,
// fragments re-read every step or kept in registers.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_lds_pattern.hip -o /tmp/mlp && /tmp/mlp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // bit 0: re-read the 18 fragments from LDS every step; bit 1: barrier per step; bit 2: two barriers + role alternation
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ in, float* out, int iters) {
  __shared__ uint4 abuf[2][18 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2;
  for (int i = threadIdx.x; i < 2 * 18 * 64; i += 512) (&abuf[0][0])[i] = in[i % (18 * 64)];
  __syncthreads();
  f32x4 acc[6];
  for (int c = 0; c < 6; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[3][6], b[3];
  for (int p = 0; p < 3; p++) {
    for (int r = 0; r < 6; r++) a[p][r] = __builtin_bit_cast(bf16x8, abuf[0][(r * 3 + p) * 64 + lane]);
    b[p] = __builtin_bit_cast(bf16x8, in[(18 + p) * 64 + lane]);
  }
  auto multiply = [&](int cur) {
    if (MODE & 1) {
#pragma unroll
      for (int r = 0; r < 6; r++) a[2][r] = __builtin_bit_cast(bf16x8, abuf[cur][(r * 3 + 2) * 64 + lane]);
#pragma unroll
      for (int r = 0; r < 6; r++) a[0][r] = __builtin_bit_cast(bf16x8, abuf[cur][(r * 3) * 64 + lane]);
#pragma unroll
      for (int r = 0; r < 6; r++) a[1][r] = __builtin_bit_cast(bf16x8, abuf[cur][(r * 3 + 1) * 64 + lane]);
      __builtin_amdgcn_sched_barrier(0);
    }
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
  };
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
    if (MODE & 4) {
      if (grp == 0) multiply(i & 1);
      __builtin_amdgcn_s_barrier();
      if (grp == 1) multiply(i & 1);
      __builtin_amdgcn_s_barrier();
    } else {
      multiply(i & 1);
      if (MODE & 2) __builtin_amdgcn_s_barrier();
    }
  }
  float s = 0.f;
  for (int c = 0; c < 6; c++) for (int r = 0; r < 4; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const uint4* in, float* out, const char* what) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512>>>(in, out, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; i++) k<MODE><<<256, 512>>>(in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double n = (double)iters * 36;      // MFMAs per wave
  printf("%-58s %.3f ms, %5.1f ns per MFMA per SIMD, %.2f PFLOP/s bf16\n", what, ms, ms * 1e6 / n / 2.0, n * 16384 * 8 * 256 / (ms * 1e-3) / 1e15);
}

int main() {
  uint4* in; float* out;
  hipMalloc(&in, 21 * 64 * sizeof(uint4)); hipMalloc(&out, 256 * 512 * sizeof(float));
  static unsigned short h[21 * 64 * 8];
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>(in, out, "registers only, no barrier");
  run<2>(in, out, "registers only, barrier per step");
  run<1>(in, out, "18 LDS fragment reads per step, no barrier");
  run<3>(in, out, "18 LDS fragment reads per step, barrier per step");
  run<5>(in, out, "LDS reads, two barriers, groups alternate (idle partner)");
  run<4>(in, out, "registers, two barriers, groups alternate (idle partner)");
  return 0;
}
