// E4: log-domain Sinkhorn with dustbin row/column (LearnableLogOptimalTransport).
//
// Replaces geotransformer/modules/sinkhorn/learnable_sinkhorn.py:13-66 (200 logsumexp launches and ~3.5 GB of
// un-fused traffic per call in the reference).  One workgroup per patch pair keeps the
// whole padded (R+1) x (C+1) score matrix in REGISTERS (each entry twice: once in a row-owner lane, once in a
// column-owner lane) and the dual vectors u, v in LDS for all iterations; HBM traffic is the algorithmic
// 4 * B * (R*C + (R+1)*(C+1)) bytes.  The arithmetic follows the reference: u = log_mu - LSE_j(Z + v),
// v = log_nu - LSE_i(Z + u), LSE = max + log(sum(exp(x - max))), masked entries = -inf_value (finite, 1e12).
#include "common.h"

namespace {

// butterfly exchange inside an aligned group of LANES (4 or 8) lanes on the DPP path (no LDS permute)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LANES>
__device__ __forceinline__ float group_max(float m) {
  if (LANES == 8) m = fmaxf(m, dpp_mov<0x141>(m));      // row_half_mirror: lane i <-> 7 - i
  m = fmaxf(m, dpp_mov<0x4E>(m));                       // quad_perm [2, 3, 0, 1]
  return fmaxf(m, dpp_mov<0xB1>(m));                    // quad_perm [1, 0, 3, 2]
}
template <int LANES>
__device__ __forceinline__ float group_sum(float s) {
  if (LANES == 8) s += dpp_mov<0x141>(s);
  s += dpp_mov<0x4E>(s);
  return s + dpp_mov<0xB1>(s);
}

constexpr int kThreads = 576;      // 9 waves: (65 rows) x 8 lanes, or (129 rows) x 4 lanes

template <int LANES, int EPL>      // lanes cooperating on one row/column (4 or 8), entries per lane
__global__ __launch_bounds__(kThreads) void sinkhorn_kernel(const float* __restrict__ scores,
                                                            const uint8_t* __restrict__ row_masks,
                                                            const uint8_t* __restrict__ col_masks,
                                                            const float* __restrict__ alpha_p, int R, int C, int iters,
                                                            float inf, float* __restrict__ out) {
  __shared__ float u[160], v[160], log_mu[160], log_nu[160];
  __shared__ float s_cnt[2];
  const int b = blockIdx.x;
  const int R1 = R + 1, C1 = C + 1;
  const int tid = threadIdx.x;
  const int owner = tid / LANES, sub = tid % LANES;      // `owner` indexes a row in the row pass, a column in the column pass
  const float alpha = alpha_p[0];
  const float* S = scores + (size_t)b * R * C;
  const uint8_t* rm = row_masks + (size_t)b * R;
  const uint8_t* cm = col_masks + (size_t)b * C;

  if (tid == 0) {
    float nr = 0.f, nc = 0.f;
    for (int i = 0; i < R; i++) nr += rm[i] ? 1.f : 0.f;
    for (int j = 0; j < C; j++) nc += cm[j] ? 1.f : 0.f;
    s_cnt[0] = nr;
    s_cnt[1] = nc;
  }
  __syncthreads();
  const float nvr = s_cnt[0], nvc = s_cnt[1];
  const float norm = -logf(nvr + nvc);
  for (int i = tid; i < R1; i += kThreads) {
    const bool masked = i < R && !rm[i];
    log_mu[i] = masked ? -inf : (i < R ? norm : logf(nvc) + norm);
    u[i] = 0.f;
  }
  for (int j = tid; j < C1; j += kThreads) {
    const bool masked = j < C && !cm[j];
    log_nu[j] = masked ? -inf : (j < C ? norm : logf(nvr) + norm);
    v[j] = 0.f;
  }

  auto zval = [&](int i, int j) -> float {
    const bool masked = (i < R && !rm[i]) || (j < C && !cm[j]);
    if (masked) return -inf;
    return (i < R && j < C) ? S[(size_t)i * C + j] : alpha;
  };
  // row-owner copy: row `owner`, columns sub, sub+LANES, ... ; column-owner copy: column `owner`, rows sub, sub+LANES, ...
  float zr[EPL], zc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) {
    const int j = sub + e * LANES;
    zr[e] = (owner < R1 && j < C1) ? zval(owner, j) : 0.f;
    const int i = sub + e * LANES;
    zc[e] = (owner < C1 && i < R1) ? zval(i, owner) : 0.f;
  }
  __syncthreads();

  for (int it = 0; it < iters; it++) {
    {  // u_i = log_mu_i - LSE_j(Z_ij + v_j)
      float t[EPL], m = -INFINITY;
#pragma unroll
      for (int e = 0; e < EPL; e++) {
        const int j = sub + e * LANES;
        t[e] = (j < C1) ? zr[e] + v[j] : -INFINITY;
        m = fmaxf(m, t[e]);
      }
      m = group_max<LANES>(m);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; e++) s += __expf(t[e] - m);
      s = group_sum<LANES>(s);
      if (sub == 0 && owner < R1) u[owner] = log_mu[owner] - (logf(s) + m);
    }
    __syncthreads();
    {  // v_j = log_nu_j - LSE_i(Z_ij + u_i)
      float t[EPL], m = -INFINITY;
#pragma unroll
      for (int e = 0; e < EPL; e++) {
        const int i = sub + e * LANES;
        t[e] = (i < R1) ? zc[e] + u[i] : -INFINITY;
        m = fmaxf(m, t[e]);
      }
      m = group_max<LANES>(m);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; e++) s += __expf(t[e] - m);
      s = group_sum<LANES>(s);
      if (sub == 0 && owner < C1) v[owner] = log_nu[owner] - (logf(s) + m);
    }
    __syncthreads();
  }

  float* O = out + (size_t)b * R1 * C1;
  if (owner < R1) {
    const float ui = u[owner];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const int j = sub + e * LANES;
      if (j < C1) O[(size_t)owner * C1 + j] = zr[e] + ui + v[j] - norm;
    }
  }
}

}  // namespace

extern "C" int se3_log_sinkhorn_fwd(const float* scores, const uint8_t* row_masks, const uint8_t* col_masks,
                                    const float* alpha, int batch, int rows, int cols, int iterations, float inf,
                                    float* out, void* stream) {
  SE3_REQUIRE(scores && row_masks && col_masks && alpha && out, SE3_ERR_INVALID_ARG, "log_sinkhorn: null pointer");
  SE3_REQUIRE(batch >= 0 && rows >= 1 && cols >= 1 && iterations >= 0, SE3_ERR_INVALID_ARG, "log_sinkhorn: bad sizes");
  const int dim = (rows > cols ? rows : cols) + 1;
  SE3_REQUIRE(dim <= 144, SE3_ERR_UNSUPPORTED, "log_sinkhorn: patches of %d x %d points exceed 143", rows, cols);
  if (batch == 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dim <= 72)
    sinkhorn_kernel<8, 9><<<batch, kThreads, 0, st>>>(scores, row_masks, col_masks, alpha, rows, cols, iterations, inf, out);
  else
    sinkhorn_kernel<4, 36><<<batch, kThreads, 0, st>>>(scores, row_masks, col_masks, alpha, rows, cols, iterations, inf, out);
  SE3_CHECK_LAUNCH("log_sinkhorn");
  return SE3_OK;
}
