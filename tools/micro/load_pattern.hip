// Does the LANE MAPPING of a 16-byte-per-lane streaming read matter?  rpe_bias_kernel reads a 16 KB tile (16 keys x 1 KB) with 16 load
// instructions of 16 rows x 64 B each; lane l = (key l & 15, 16-byte piece l >> 4), so the four lanes that read one row's 64 contiguous
// bytes are 16 lanes apart.  Variant 1 puts them side by side (lane l = (row l >> 2, piece l & 3)); variant 2 is the plain float4 stream
// (64 lanes x 16 B contiguous).  Same bytes, same number of instructions in flight, one resident round of 4-wave workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/load_pattern.hip -o tools/micro/load_pattern && tools/micro/load_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void stream_kernel(const float4* __restrict__ src, long long tiles, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 b[16];
  // a tile = 16 rows x 64 float4 (1 KB per row); chunk t of a row = float4 4 t .. 4 t + 3
  auto addr = [&](long long tile, int t) -> const float4* {
    if (VARIANT == 0) return src + tile * 1024 + (lane & 15) * 64 + 4 * t + (lane >> 4);          // row = lane & 15, piece = lane >> 4
    if (VARIANT == 1) return src + tile * 1024 + (lane >> 2) * 64 + 4 * t + (lane & 3);           // row = lane >> 2, piece = lane & 3
    return src + tile * 1024 + t * 64 + lane;                                                       // plain stream
  };
  long long tile = wave;
  if (tile < tiles)
#pragma unroll
    for (int t = 0; t < 16; t++) b[t] = *addr(tile, t);
  for (; tile < tiles; tile += nwaves) {
    const long long nxt = tile + nwaves < tiles ? tile + nwaves : tile;
#pragma unroll
    for (int t = 0; t < 16; t++) {
      acc.x += b[t].x; acc.y += b[t].y; acc.z += b[t].z; acc.w += b[t].w;
      b[t] = *addr(nxt, t);                                   // rotating prefetch: one tile per wave in flight
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

int main() {
  const long long bytes = 2ll << 30, tiles = bytes / 16384;
  float4* src;
  float* sink;
  hipMalloc(&src, bytes);
  hipMalloc(&sink, 4);
  hipMemset(src, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs_per_cu = 2; wgs_per_cu <= 3; wgs_per_cu++)
    for (int v = 0; v < 3; v++) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        const int grid = 256 * wgs_per_cu;
        if (v == 0) stream_kernel<0><<<grid, 256>>>(src, tiles, sink);
        else if (v == 1) stream_kernel<1><<<grid, 256>>>(src, tiles, sink);
        else stream_kernel<2><<<grid, 256>>>(src, tiles, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("%d workgroups per CU, variant %d (%s): %.1f us  %.2f TB/s\n", wgs_per_cu, v,
             v == 0 ? "rows by lane & 15: a row's four 16-B pieces 16 lanes apart (rpe_bias_kernel)" : v == 1 ? "rows by lane >> 2: four adjacent lanes read 64 contiguous bytes" : "plain float4 stream",
             best * 1e3f, bytes / (best * 1e-3) / 1e12);
    }
  return 0;
}
