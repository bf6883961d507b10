// B2 / D: dense layers y = x W^T [+ b] of the backbone (UnaryBlockEPN.mlp, blocks_epn.py:639-665) and of the transformer
// (geotransformer/modules/transformer/*: proj_q / proj_k / proj_p / FFN linears) on the f16 matrix cores at f32 accuracy.
//
// The library f32 GEMMs of these layers run at 100-120 TFLOP/s: they are bound by the f32 MFMA rate (157 TFLOP/s), 1/16 of the f16 rate.
// Here both operands are f16 hi + lo pieces (x = hi + lo to 2^-22 |x|; the weights pre-split once per weight version and scaled by a power
// of two so that their lo pieces stay normal numbers, the activations split in registers on their way into LDS) and the three products
// hi hi + hi lo + lo hi accumulate in f32 with v_mfma_f32_32x32x16_f16: 3 MFMAs of the 16x faster kind per f32 MFMA's worth of work, error
// 2^-22 per term (below the f32 GEMM's own accumulation error; tests/test_gpu_ops.py::test_linear_f16_split_has_f32_accuracy).
//
// Workgroup tile 128 rows x BN columns (BN 128: 2 x 2 waves of 64 x 64; BN 64: 4 x 1 waves of 32 x 64), K in steps of 32:
//   * A: every thread loads 16 consecutive floats of one row (64 B), splits them and stores the two 32-byte runs into the LDS image
//     [piece][row][32 k] (row stride 80 B: the 16 rows of a ds_read_b128 lane group fall into different bank quads); double buffered, the
//     loads of step k + 1 are in flight while step k multiplies; one barrier per step;
//   * B: weight fragments in lane order (se3_linear_split_weights_f16), straight from L1 / L2, one K16 sub-step ahead;
//   * epilogue: scale, bias, optional ReLU, 128-byte row segments.
#include "common.h"

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kHeaderB = 256;                 // weight-piece buffer: [header: 1 / scale, max |W| bits][fragments]
constexpr int kBM = 128, kBK = 32;
constexpr int kRowB = 80;                     // bytes per (piece, row) of the A image: 32 f16 + 16 B pad
constexpr int kPieceB = kBM * kRowB, kBufB = 2 * kPieceB;

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

template <int BN>
__global__ __launch_bounds__(256) void linear_f16_kernel(const float* __restrict__ x, int64_t M, int K, int64_t x_rs, const u32x4* __restrict__ Wf,
                                                         const float* __restrict__ hdr, const float* __restrict__ bias, int N, int NCT,
                                                         float* __restrict__ out, int64_t out_rs, int relu) {
  constexpr int RT = BN == 128 ? 2 : 1;               // 32-row tiles per wave
  constexpr int CT = 2;                               // 32-column tiles per wave
  __shared__ __align__(16) unsigned char lds[2 * kBufB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0;
  const int ncb = NCT / (BN / 32);
  const int cb = blockIdx.x % ncb;
  const int64_t row0 = (int64_t)(blockIdx.x / ncb) * kBM;
  const int i32 = lane & 31, h = lane >> 5;
  // A staging: thread -> (row, half): 16 consecutive floats
  const int ar = tid >> 1, ah = tid & 1;
  const float* xrow = x + (row0 + ar < M ? row0 + ar : M - 1) * x_rs + 16 * ah;
  const int a_store = ar * kRowB + ah * 32;
  const int a_read = (wm * (RT * 32) + i32) * kRowB + h * 16;          // + rt * 32 * kRowB + ks * 32 + piece * kPieceB
  const int ct0 = cb * (BN / 32) + wn * CT;
  const u32x4* wbase = Wf + (int64_t)ct0 * 2 * 64 + lane;
  const int64_t wstep = (int64_t)NCT * 2 * 64;                          // u32x4 per K16-step
  f32x16 acc[RT][CT];
#pragma unroll
  for (int r = 0; r < RT; r++)
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[r][c][v] = 0.f;
  const int nk = K / kBK;
  const int64_t last16 = (int64_t)2 * nk - 1;
  f32x4 an[4];
#pragma unroll
  for (int q = 0; q < 4; q++) an[q] = *reinterpret_cast<const f32x4*>(xrow + 4 * q);
  u32x4 bq[2][CT][2];                                                   // weight fragments of the next two K16 sub-steps
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int c = 0; c < CT; c++) {
      const int64_t g = j < last16 ? j : last16;
      bq[j][c][0] = wbase[g * wstep + c * 128];
      bq[j][c][1] = wbase[g * wstep + c * 128 + 64];
    }
  auto stage = [&](int buf) {                                           // an -> f16 hi / lo -> LDS image `buf`
    f16x8 hi[2], lo[2];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float v = an[q][e];
        const _Float16 hv = (_Float16)v;
        hi[q >> 1][(q & 1) * 4 + e] = hv;
        lo[q >> 1][(q & 1) * 4 + e] = (_Float16)(v - (float)hv);
      }
    unsigned char* dst = lds + buf * kBufB + a_store;
    *reinterpret_cast<f16x8*>(dst) = hi[0];
    *reinterpret_cast<f16x8*>(dst + 16) = hi[1];
    *reinterpret_cast<f16x8*>(dst + kPieceB) = lo[0];
    *reinterpret_cast<f16x8*>(dst + kPieceB + 16) = lo[1];
  };
  stage(0);
  __syncthreads();
  for (int kk = 0; kk < nk; kk++) {
    {
      const int kn = kk + 1 < nk ? kk + 1 : kk;                          // unconditional (clamped) so that the compiler can count the requests
#pragma unroll
      for (int q = 0; q < 4; q++) an[q] = *reinterpret_cast<const f32x4*>(xrow + (int64_t)kn * kBK + 4 * q);
    }
    const unsigned char* img = lds + (kk & 1) * kBufB + a_read;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      f16x8 av[RT][2];
#pragma unroll
      for (int r = 0; r < RT; r++) {
        av[r][0] = *reinterpret_cast<const f16x8*>(img + r * 32 * kRowB + ks * 32);
        av[r][1] = *reinterpret_cast<const f16x8*>(img + r * 32 * kRowB + ks * 32 + kPieceB);
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const f16x8 b0 = __builtin_bit_cast(f16x8, bq[ks][c][0]);
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[r][1], b0, acc[r][c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const f16x8 b1 = __builtin_bit_cast(f16x8, bq[ks][c][1]);
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[r][0], b1, acc[r][c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const f16x8 b0 = __builtin_bit_cast(f16x8, bq[ks][c][0]);
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[r][0], b0, acc[r][c], 0, 0, 0);
      }
      {
        int64_t g = (int64_t)2 * kk + ks + 2;
        g = g < last16 ? g : last16;
#pragma unroll
        for (int c = 0; c < CT; c++) {
          bq[ks][c][0] = wbase[g * wstep + c * 128];
          bq[ks][c][1] = wbase[g * wstep + c * 128 + 64];
        }
      }
    }
    stage((kk + 1) & 1);
    __syncthreads();
  }
  const float inv_scale = hdr[0];
#pragma unroll
  for (int c = 0; c < CT; c++) {
    const int col = (ct0 + c) * 32 + i32;
    const float bv = (bias != nullptr && col < N) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int64_t row = row0 + wm * (RT * 32) + r * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        float val = acc[r][c][v] * inv_scale + bv;
        if (relu) val = fmaxf(val, 0.f);
        if (row < M && col < N) out[row * out_rs + col] = val;
      }
  }
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

extern "C" int se3_linear_f16(const float* x, int64_t rows, int in_features, int64_t x_row_stride, const void* weight_pieces,
                              const float* bias, int out_features, int apply_relu, float* out, int64_t out_row_stride, void* stream) {
  SE3_REQUIRE(x && weight_pieces && out, SE3_ERR_INVALID_ARG, "linear_f16: null pointer");
  SE3_REQUIRE(in_features > 0 && in_features % 32 == 0 && out_features > 0, SE3_ERR_UNSUPPORTED,
              "linear_f16: in_features %d must be a multiple of 32", in_features);
  SE3_REQUIRE(x_row_stride >= in_features && x_row_stride % 4 == 0 && out_row_stride >= out_features && ((uintptr_t)x & 15) == 0,
              SE3_ERR_INVALID_ARG, "linear_f16: rows must be 16-byte aligned (strides %lld, %lld)", (long long)x_row_stride,
              (long long)out_row_stride);
  if (rows == 0) return SE3_OK;
  const int NCT = (out_features + 63) / 64 * 2;
  const float* hdr = static_cast<const float*>(weight_pieces);
  const u32x4* Wf = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces) + kHeaderB);
  const int64_t rt = se3_cdiv(rows, kBM);
  hipStream_t st = (hipStream_t)stream;
  if (NCT % 4 == 0)
    linear_f16_kernel<128><<<(unsigned)(rt * (NCT / 4)), 256, 0, st>>>(x, rows, in_features, x_row_stride, Wf, hdr, bias, out_features, NCT, out,
                                                                      out_row_stride, apply_relu);
  else
    linear_f16_kernel<64><<<(unsigned)(rt * (NCT / 2)), 256, 0, st>>>(x, rows, in_features, x_row_stride, Wf, hdr, bias, out_features, NCT, out,
                                                                     out_row_stride, apply_relu);
  SE3_CHECK_LAUNCH("linear_f16");
  return SE3_OK;
}
