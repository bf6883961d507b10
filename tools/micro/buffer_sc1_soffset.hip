// Does a buffer store / load with sc1 (agent scope) honour soffset?  hipcc --offload-arch=gfx950 -O3 buffer_sc1_soffset.hip -o buffer_sc1_soffset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int AUX>
__global__ void k(float* buf, float* out, int row_b) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 16 * row_b, 0x00020000);
  const int lane = threadIdx.x;
#pragma unroll
  for (int v = 0; v < 16; v++)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(v * 1000 + lane)), rs, lane * 4, v * row_b, AUX);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int v = 0; v < 16; v++)
    out[v * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, v * row_b, AUX));
}
int main() {
  float *buf, *out;
  hipMalloc(&buf, 16 * 256 * 4);
  hipMalloc(&out, 16 * 64 * 4);
  for (int aux : {0, 16}) {
    hipMemset(buf, 0, 16 * 256 * 4);
    if (aux) k<16><<<1, 64>>>(buf, out, 256); else k<0><<<1, 64>>>(buf, out, 256);
    std::vector<float> h(16 * 64), b(16 * 64);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), buf, b.size() * 4, hipMemcpyDeviceToHost);
    int bad_load = 0, bad_mem = 0;
    for (int v = 0; v < 16; v++)
      for (int l = 0; l < 64; l++) {
        bad_load += h[v * 64 + l] != (float)(v * 1000 + l);
        bad_mem += b[v * 64 + l] != (float)(v * 1000 + l);
      }
    printf("aux %2d: loads wrong %d, memory wrong %d (row 1 lane 0: load %.0f mem %.0f)\n", aux, bad_load, bad_mem, h[64], b[64]);
  }
  return 0;
}
