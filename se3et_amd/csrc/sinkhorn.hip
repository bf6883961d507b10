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

// FAST (the default; late round 5 -- the kernel is bound by its vector instructions: 87 % of the issue slots of 9 waves x 74 instructions per
// half-iteration): the same iteration in base 2 -- Z, u, v, log mu, log nu times log2(e) once, so that an exponential is ONE v_exp_f32 and the
// logarithm ONE v_log_f32 (natural-base expf = multiply + v_exp_f32, logf = a ten-instruction sequence), converted back in the epilogue --,
// entries beyond the matrix hold -inf from the start instead of being selected in every pass, and the shift of a row's / column's
// logsumexp is the value of the SAME logsumexp one iteration earlier instead of the exact maximum (the sum then sits near 1; a sum outside
// [2^-60, 2^60] -- the duals moved far, an overflow, the first iteration -- sends the whole wave through the exact form: the result is the
// logsumexp either way, to rounding).  !FAST: the reference's order of operations, kept for A/B runs (se3_debug_set_sinkhorn_variant).
template <int LANES, int EPL, bool FAST>      // lanes cooperating on one row/column (4 or 8), entries per lane
__global__ __launch_bounds__(kThreads) void sinkhorn_kernel(const float* __restrict__ scores,
                                                            const uint8_t* __restrict__ row_masks,
                                                            const uint8_t* __restrict__ col_masks,
                                                            const float* __restrict__ alpha_p, int R, int C, int iters,
                                                            float inf, float* __restrict__ out) {
  __shared__ float u[160], v[160], log_mu[160], log_nu[160];
  constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
  const float unit = FAST ? kLog2e : 1.f;                 // the dual variables and Z live in units of 1 / unit nats
  const int b = blockIdx.x;
  const int R1 = R + 1, C1 = C + 1;
  const int tid = threadIdx.x;
  const int owner = tid / LANES, sub = tid % LANES;      // `owner` indexes a row in the row pass, a column in the column pass
  const float alpha = alpha_p[0];
  const float* S = scores + (size_t)b * R * C;
  const uint8_t* rm = row_masks + (size_t)b * R;
  const uint8_t* cm = col_masks + (size_t)b * C;

  // valid rows / columns, counted by all threads (one thread walking the mask bytes was a chain of global loads in front of everything: 4 %)
  const float nvr = (float)__syncthreads_count(tid < R && rm[tid < R ? tid : 0] != 0);
  const float nvc = (float)__syncthreads_count(tid < C && cm[tid < C ? tid : 0] != 0);
  const float norm = -logf(nvr + nvc);
  for (int i = tid; i < 160; i += kThreads) {
    const bool masked = i < R && !rm[i];
    log_mu[i] = i < R1 ? (masked ? -inf : (i < R ? norm : logf(nvc) + norm)) * unit : 0.f;
    u[i] = 0.f;                                           // (also beyond the matrix: FAST reads those entries and adds them to -inf)
  }
  for (int j = tid; j < 160; j += kThreads) {
    const bool masked = j < C && !cm[j];
    log_nu[j] = j < C1 ? (masked ? -inf : (j < C ? norm : logf(nvr) + norm)) * unit : 0.f;
    v[j] = 0.f;
  }

  auto zval = [&](int i, int j) -> float {
    const bool masked = (i < R && !rm[i]) || (j < C && !cm[j]);
    if (masked) return -inf * unit;
    return ((i < R && j < C) ? S[(size_t)i * C + j] : alpha) * unit;
  };
  // row-owner copy: row `owner`, columns sub, sub+LANES, ... ; column-owner copy: column `owner`, rows sub, sub+LANES, ...
  const float outside = FAST ? -INFINITY : 0.f;
  float zr[EPL], zc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) {
    const int j = sub + e * LANES;
    zr[e] = owner < R1 ? (j < C1 ? zval(owner, j) : outside) : 0.f;
    const int i = sub + e * LANES;
    zc[e] = owner < C1 ? (i < R1 ? zval(i, owner) : outside) : 0.f;
  }
  __syncthreads();

  if (FAST) {
    // one half-iteration: dual[owner] = log_m[owner] - LSE2_e(z[e] + other[sub + e LANES]); lse: this owner's logsumexp of the previous iteration
    auto half = [&](const float (&z)[EPL], const float* other, const float* log_m, float* dual, int n_own, float& lse, bool first) {
      float t[EPL];
#pragma unroll
      for (int e = 0; e < EPL; e++) t[e] = z[e] + other[sub + e * LANES];
      float s = 0.f;
      if (!first) {
#pragma unroll
        for (int e = 0; e < EPL; e++) s += __builtin_amdgcn_exp2f(t[e] - lse);
        s = group_sum<LANES>(s);
      }
      if (__any(owner < n_own && !(s > 8.67e-19f && s < 1.15e18f))) {       // (wave-uniform; NaN fails both) the exact form
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < EPL; e++) m = fmaxf(m, t[e]);
        m = group_max<LANES>(m);
        s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; e++) s += __builtin_amdgcn_exp2f(t[e] - m);
        s = group_sum<LANES>(s);
        lse = m;
      }
      lse += __builtin_amdgcn_logf(s);
      if (sub == 0 && owner < n_own) dual[owner] = log_m[owner] - lse;
    };
    float lse_r = 0.f, lse_c = 0.f;
    for (int it = 0; it < iters; it++) {
      half(zr, v, log_mu, u, R1, lse_r, it == 0);
      __syncthreads();
      half(zc, u, log_nu, v, C1, lse_c, it == 0);
      __syncthreads();
    }
  } else {
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
  }

  float* O = out + (size_t)b * R1 * C1;
  if (owner < R1) {
    const float ui = u[owner];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const int j = sub + e * LANES;
      if (j < C1) O[(size_t)owner * C1 + j] = FAST ? (zr[e] + ui + v[j]) * kLn2 - norm : zr[e] + ui + v[j] - norm;
    }
  }
}

// ---- backward (training step) ------------------------------------------------------------------------------------------------------------
// Reverse-mode differentiation of the iteration above, one workgroup per patch pair, same thread mapping (a row-owner and a column-owner
// register copy of Z).  Phase 1 repeats the forward iterations and keeps every u_t, v_t in LDS (iterations x (R1 + C1) floats); phase 2
// walks them backwards:   v_t = log_nu - LSE_i(Z + u_t):  P2 = exp((Z_ij + u_t[i]) - (log_nu[j] - v_t[j])),  dZ -= dv[j] P2,  du[i] -= sum_j dv[j] P2
//                         u_t = log_mu - LSE_j(Z + v_{t-1}): P1 = exp((Z_ij + v_{t-1}[j]) - (log_mu[i] - u_t[i])), dZ -= du[i] P1,  dv[j]  = -sum_i du[i] P1
// (the softmax weights from the saved LSE, as torch.logsumexp's backward).  The row pass accumulates its dZ terms in the row-owner copy, the
// column pass in the column-owner copy; they meet in the output.  grad_scores (B, R, C) (masked entries 0), grad_alpha_partial (B): the
// bin entries' sum per patch pair (the caller adds them up).
template <int LANES, int EPL>
__global__ __launch_bounds__(kThreads) void sinkhorn_bwd_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ row_masks,
                                                                const uint8_t* __restrict__ col_masks, const float* __restrict__ alpha_p,
                                                                const float* __restrict__ grad_out, int R, int C, int iters, float inf,
                                                                float* __restrict__ grad_scores, float* __restrict__ grad_alpha_partial) {
  extern __shared__ float hist[];                        // u_t: [iters][R1], then v_t: [iters][C1]
  __shared__ float u[160], v[160], log_mu[160], log_nu[160], du[160], dv[160];
  __shared__ float s_alpha[kThreads / 64];
  const int b = blockIdx.x;
  const int R1 = R + 1, C1 = C + 1;
  float* uh = hist;
  float* vh = hist + (size_t)iters * R1;
  const int tid = threadIdx.x;
  const int owner = tid / LANES, sub = tid % LANES;
  const float alpha = alpha_p[0];
  const float* S = scores + (size_t)b * R * C;
  const float* GO = grad_out + (size_t)b * R1 * C1;
  const uint8_t* rm = row_masks + (size_t)b * R;
  const uint8_t* cm = col_masks + (size_t)b * C;
  // valid rows / columns, counted by all threads (one thread walking the mask bytes was a chain of global loads in front of everything: 4 %)
  const float nvr = (float)__syncthreads_count(tid < R && rm[tid < R ? tid : 0] != 0);
  const float nvc = (float)__syncthreads_count(tid < C && cm[tid < C ? tid : 0] != 0);
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
  auto is_masked = [&](int i, int j) -> bool { return (i < R && !rm[i]) || (j < C && !cm[j]); };
  auto zval = [&](int i, int j) -> float {
    if (is_masked(i, j)) return -inf;
    return (i < R && j < C) ? S[(size_t)i * C + j] : alpha;
  };
  float zr[EPL], zc[EPL], gr[EPL], gc[EPL];              // Z and the accumulated dZ terms in both ownerships
#pragma unroll
  for (int e = 0; e < EPL; e++) {
    const int j = sub + e * LANES;
    zr[e] = (owner < R1 && j < C1) ? zval(owner, j) : 0.f;
    gr[e] = (owner < R1 && j < C1) ? GO[(size_t)owner * C1 + j] : 0.f;       // starts as dL/dout
    const int i = sub + e * LANES;
    zc[e] = (owner < C1 && i < R1) ? zval(i, owner) : 0.f;
    gc[e] = 0.f;
  }
  __syncthreads();
  // phase 1: the forward iterations, history kept
  for (int it = 0; it < iters; it++) {
    {
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
      if (sub == 0 && owner < R1) {
        const float un = log_mu[owner] - (logf(s) + m);
        u[owner] = un;
        uh[(size_t)it * R1 + owner] = un;
      }
    }
    __syncthreads();
    {
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
      if (sub == 0 && owner < C1) {
        const float vn = log_nu[owner] - (logf(s) + m);
        v[owner] = vn;
        vh[(size_t)it * C1 + owner] = vn;
      }
    }
    __syncthreads();
  }
  // phase 2.  out = Z + u_T + v_T - norm: du = row sums, dv = column sums of dL/dout
  {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; e++) s += gr[e];
    s = group_sum<LANES>(s);
    if (sub == 0 && owner < R1) du[owner] = s;
    float c = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const int i = sub + e * LANES;
      c += (owner < C1 && i < R1) ? GO[(size_t)i * C1 + owner] : 0.f;
    }
    c = group_sum<LANES>(c);
    if (sub == 0 && owner < C1) dv[owner] = c;
  }
  __syncthreads();
  for (int it = iters - 1; it >= 0; it--) {
    const float* ut = uh + (size_t)it * R1;
    const float* vt = vh + (size_t)it * C1;
    {  // through v_t (row-owner): needs dv, adds to du
      float acc = 0.f;
      const float ui = owner < R1 ? ut[owner] : 0.f;
#pragma unroll
      for (int e = 0; e < EPL; e++) {
        const int j = sub + e * LANES;
        if (owner < R1 && j < C1) {
          const float p = __expf((zr[e] + ui) - (log_nu[j] - vt[j]));
          const float c = dv[j] * p;
          gr[e] -= c;
          acc += c;
        }
      }
      acc = group_sum<LANES>(acc);
      if (sub == 0 && owner < R1) du[owner] -= acc;
    }
    __syncthreads();
    {  // through u_t (column-owner): needs du, replaces dv
      float acc = 0.f;
      const float vj = (it > 0 && owner < C1) ? vh[(size_t)(it - 1) * C1 + owner] : 0.f;
#pragma unroll
      for (int e = 0; e < EPL; e++) {
        const int i = sub + e * LANES;
        if (owner < C1 && i < R1) {
          const float p = __expf((zc[e] + vj) - (log_mu[i] - ut[i]));
          const float c = du[i] * p;
          gc[e] -= c;
          acc += c;
        }
      }
      acc = group_sum<LANES>(acc);
      __syncthreads();                                 // every du[i] read before it is cleared
      if (sub == 0 && owner < C1) dv[owner] = -acc;
      if (tid < R1) du[tid] = 0.f;
      if (R1 > kThreads) for (int i = tid + kThreads; i < R1; i += kThreads) du[i] = 0.f;
    }
    __syncthreads();
  }
  // dZ = row-owner terms + column-owner terms: the row owners store, then the column owners add (same workgroup, global memory)
  float* GS = grad_scores + (size_t)b * R * C;
  float bin = 0.f;
  if (owner < R1) {
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const int j = sub + e * LANES;
      if (j < C1 && !is_masked(owner, j)) {
        if (owner < R && j < C) GS[(size_t)owner * C + j] = gr[e];
        else bin += gr[e];
      } else if (owner < R && j < C) {
        GS[(size_t)owner * C + j] = 0.f;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  if (owner < C1) {
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const int i = sub + e * LANES;
      if (i < R1 && !is_masked(i, owner)) {
        if (i < R && owner < C) GS[(size_t)i * C + owner] += gc[e];
        else bin += gc[e];
      }
    }
  }
  // the bin entries' gradient -> alpha
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bin += __shfl_xor(bin, o);
  if ((tid & 63) == 0) s_alpha[tid >> 6] = bin;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < kThreads / 64; w++) t += s_alpha[w];
    grad_alpha_partial[b] = t;
  }
}

}  // namespace

static int g_sinkhorn_variant = 0;          // 0: base-2 iteration with carried shifts (default); 1: the reference's order of operations
extern "C" void se3_debug_set_sinkhorn_variant(int variant) { g_sinkhorn_variant = variant; }

extern "C" int se3_log_sinkhorn_fwd(const float* scores, const uint8_t* row_masks, const uint8_t* col_masks,
                                    const float* alpha, int batch, int rows, int cols, int iterations, float inf,
                                    float* out, void* stream) {
  SE3_REQUIRE(scores && row_masks && col_masks && alpha && out, SE3_ERR_INVALID_ARG, "log_sinkhorn: null pointer");
  SE3_REQUIRE(batch >= 0 && rows >= 1 && cols >= 1 && iterations >= 0, SE3_ERR_INVALID_ARG, "log_sinkhorn: bad sizes");
  const int dim = (rows > cols ? rows : cols) + 1;
  SE3_REQUIRE(dim <= 144, SE3_ERR_UNSUPPORTED, "log_sinkhorn: patches of %d x %d points exceed 143", rows, cols);
  if (batch == 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dim <= 72) {
    if (g_sinkhorn_variant == 0) sinkhorn_kernel<8, 9, true><<<batch, kThreads, 0, st>>>(scores, row_masks, col_masks, alpha, rows, cols, iterations, inf, out);
    else sinkhorn_kernel<8, 9, false><<<batch, kThreads, 0, st>>>(scores, row_masks, col_masks, alpha, rows, cols, iterations, inf, out);
  } else {
    if (g_sinkhorn_variant == 0) sinkhorn_kernel<4, 36, true><<<batch, kThreads, 0, st>>>(scores, row_masks, col_masks, alpha, rows, cols, iterations, inf, out);
    else sinkhorn_kernel<4, 36, false><<<batch, kThreads, 0, st>>>(scores, row_masks, col_masks, alpha, rows, cols, iterations, inf, out);
  }
  SE3_CHECK_LAUNCH("log_sinkhorn");
  return SE3_OK;
}

// Backward of se3_log_sinkhorn_fwd (autograd of learnable_sinkhorn.py:13-66 in the training step): grad_out (B, R+1, C+1) -> grad_scores
// (B, R, C) and grad_alpha_partial (B) (the caller sums it).  Needs iterations * (R + C + 2) floats of LDS (<= 150 KB).
extern "C" int se3_log_sinkhorn_bwd(const float* scores, const uint8_t* row_masks, const uint8_t* col_masks, const float* alpha,
                                    const float* grad_out, int batch, int rows, int cols, int iterations, float inf, float* grad_scores,
                                    float* grad_alpha_partial, void* stream) {
  SE3_REQUIRE(scores && row_masks && col_masks && alpha && grad_out && grad_scores && grad_alpha_partial, SE3_ERR_INVALID_ARG,
              "log_sinkhorn_bwd: null pointer");
  SE3_REQUIRE(batch >= 0 && rows >= 1 && cols >= 1 && iterations >= 0, SE3_ERR_INVALID_ARG, "log_sinkhorn_bwd: bad sizes");
  const int dim = (rows > cols ? rows : cols) + 1;
  SE3_REQUIRE(dim <= 144, SE3_ERR_UNSUPPORTED, "log_sinkhorn_bwd: patches of %d x %d points exceed 143", rows, cols);
  const size_t lds = (size_t)iterations * (rows + cols + 2) * sizeof(float);
  SE3_REQUIRE(lds <= 150 * 1024, SE3_ERR_UNSUPPORTED, "log_sinkhorn_bwd: %d iterations of %d + %d duals do not fit in LDS", iterations,
              rows + 1, cols + 1);
  if (batch == 0) return SE3_OK;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_bwd_kernel<8, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_bwd_kernel<4, 36>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dim <= 72)
    sinkhorn_bwd_kernel<8, 9><<<batch, kThreads, lds, st>>>(scores, row_masks, col_masks, alpha, grad_out, rows, cols, iterations, inf,
                                                          grad_scores, grad_alpha_partial);
  else
    sinkhorn_bwd_kernel<4, 36><<<batch, kThreads, lds, st>>>(scores, row_masks, col_masks, alpha, grad_out, rows, cols, iterations, inf,
                                                           grad_scores, grad_alpha_partial);
  SE3_CHECK_LAUNCH("log_sinkhorn_bwd");
  return SE3_OK;
}
