// B2 / D: the WEIGHT side of the dense layers y = x W^T [+ b] of the backbone (UnaryBlockEPN.mlp, blocks_epn.py:639-665) and of the
// transformer (geotransformer/modules/transformer/*: proj_q / proj_k / proj_p / FFN linears) on the f16 matrix cores at f32 accuracy.
//
// The library f32 GEMMs of these layers run at 100-120 TFLOP/s: they are bound by the f32 MFMA rate (157 TFLOP/s), 1/16 of the f16 rate.
// Both operands are multiplied as f16 hi + lo pieces (x = hi + lo to 2^-22 |x|), three products hi hi + hi lo + lo hi in f32 on
// v_mfma_f32_32x32x16_f16.  This file splits the weights once per weight version: scaled by a power of two so that their lo pieces stay
// normal numbers, stored as MFMA B fragments in lane order (se3_linear_split_weights_f16).  The activations are split in registers on their
// way into LDS, scaled per group of 8 rows, by the one kernel that multiplies them: csrc/dense_norm.hip (se3_linear_stream*, se3_dense_*).
#include "common.h"

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kHeaderB = 256;                 // weight-piece buffer: [header: 1 / scale, max |W| bits][fragments]

__global__ void linear_wmax_kernel(const float* __restrict__ W, int64_t n, unsigned* __restrict__ hdr) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(W[i]));
  m = se3_wave_max(m);
  if (se3_lane() == 0) atomicMax(hdr + 1, __float_as_uint(m));      // non-negative floats order like their bit patterns
}

__device__ __forceinline__ float weight_scale(unsigned max_bits) {   // power of two bringing max |W| into [2^12, 2^13)
  const int e = (int)((max_bits >> 23) & 0xff);
  if (e == 0 || e == 0xff) return 1.f;
  return __uint_as_float((unsigned)(127 + 12 - (e - 127)) << 23);
}

// W (N, K) row-major f32 -> fragments [K16-step][column tile of 32 (padded to Np / 32)][piece][lane] x 16 B:
// lane (n = l & 31, h = l >> 5) holds W[ct * 32 + n][16 step + 8 h .. + 7]
__global__ void linear_split_weights_kernel(const float* __restrict__ W, int N, int K, int NCT, unsigned* __restrict__ hdr, u32x4* __restrict__ Wf) {
  const int64_t frag = blockIdx.x;                    // (K16-step, column tile)
  const int ct = (int)(frag % NCT), st = (int)(frag / NCT);
  const int lane = threadIdx.x, n = ct * 32 + (lane & 31), k0 = 16 * st + 8 * (lane >> 5);
  const float scale = weight_scale(hdr[1]);
  if (frag == 0 && lane == 0) reinterpret_cast<float*>(hdr)[0] = 1.f / scale;
  f16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const float w = n < N ? W[(int64_t)n * K + k0 + j] * scale : 0.f;
    hi[j] = (_Float16)w;
    lo[j] = (_Float16)(w - (float)hi[j]);
  }
  u32x4* dst = Wf + frag * 2 * 64 + lane;
  dst[0] = __builtin_bit_cast(u32x4, hi);
  dst[64] = __builtin_bit_cast(u32x4, lo);
}

}  // namespace

extern "C" size_t se3_linear_weight_pieces_bytes(int out_features, int in_features) {
  if (out_features <= 0 || in_features <= 0 || in_features % 32) return 0;
  const size_t nct = (size_t)(out_features + 63) / 64 * 2;
  return kHeaderB + (size_t)(in_features / 16) * nct * 2 * 64 * sizeof(u32x4);
}

extern "C" int se3_linear_split_weights_f16(const float* weight, int out_features, int in_features, void* pieces, void* stream) {
  SE3_REQUIRE(weight && pieces, SE3_ERR_INVALID_ARG, "linear_split_weights_f16: null pointer");
  SE3_REQUIRE(out_features > 0 && in_features > 0 && in_features % 32 == 0, SE3_ERR_UNSUPPORTED,
              "linear_split_weights_f16: in_features %d must be a multiple of 32", in_features);
  hipStream_t st = (hipStream_t)stream;
  unsigned* hdr = static_cast<unsigned*>(pieces);
  if (hipMemsetAsync(hdr, 0, kHeaderB, st) != hipSuccess) {
    se3_set_error("linear_split_weights_f16: memset failed");
    return SE3_ERR_LAUNCH;
  }
  const int64_t n = (int64_t)out_features * in_features;
  linear_wmax_kernel<<<(unsigned)(n / 4096 < 1 ? 1 : (n / 4096 > 1024 ? 1024 : n / 4096)), 256, 0, st>>>(weight, n, hdr);
  const int NCT = (out_features + 63) / 64 * 2;
  const int64_t frags = (int64_t)(in_features / 16) * NCT;
  linear_split_weights_kernel<<<(unsigned)frags, 64, 0, st>>>(weight, out_features, in_features, NCT, hdr,
                                                             reinterpret_cast<u32x4*>(static_cast<unsigned char*>(pieces) + kHeaderB));
  SE3_CHECK_LAUNCH("linear_split_weights_f16");
  return SE3_OK;
}

// (rounds 3-4: a tile-at-a-time kernel of its own; since round 5 the streaming kernel of csrc/dense_norm.hip -- faster on every shape of the
// path (profiles/r04_linear_stream_shapes.txt) and the one that carries the row scales of the activation split -- serves this entry too)
extern "C" int se3_linear_f16(const float* x, int64_t rows, int in_features, int64_t x_row_stride, const void* weight_pieces,
                              const float* bias, int out_features, int apply_relu, float* out, int64_t out_row_stride, void* stream) {
  SE3_REQUIRE(x_row_stride >= in_features, SE3_ERR_INVALID_ARG, "linear_f16: row stride %lld below in_features", (long long)x_row_stride);
  return se3_linear_stream(x, rows, in_features, x_row_stride, weight_pieces, bias, out_features, apply_relu, out, out_row_stride, stream);
}
