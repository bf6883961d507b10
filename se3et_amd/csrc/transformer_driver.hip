// D7 / D8: the coarse transformer of a batch of pairs issued from C -- every launch of RPEConditionalTransformer.forward
// (geotransformer/modules/transformer/conditional_transformer.py:251-390) and of the layers it schedules (rpe_transformer.py:134-194,
// vanilla_transformer.py:872-946, output_layer.py:7-47), followed by GeometricTransformer's out_proj (geotransformer.py:310-317), on the
// caller's stream with caller-owned workspaces.  No Python between the ~130 launches of a forward (one pair per
// forward spent 3.2 ms of host time on 484 Python-issued launches; three batches in flight fought for the interpreter).
//
// The function is a HOST-side schedule only: every launch goes through the C-ABI entry points of this library (se3_linear_stream,
// se3_rpe_self_attention_stack_fwd, se3_attention_stack_fwd, se3_cross_eq_stack_x6_fwd, se3_add_layer_norm_fwd, ...), in the order and
// with the operands of se3et_amd/batched.py::transformer_pairs, whose results it reproduces bit for bit.  Layout: packed rows, the refs of
// all pairs first (rows0 rows), then the srcs; every cloud starts at a multiple of 32 rows.
#include <cstddef>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "../../include/se3et_hip.h"
#include "common.h"
#include <cstdlib>

namespace {

struct Arena {
  unsigned char* base;
  size_t size, used;
  bool ok;
  void* take(size_t bytes) {
    const size_t at = (used + 255) & ~(size_t)255;
    if (at + bytes > size) {
      ok = false;
      return base;      // (reported by the caller; never dereferenced by a launch: the driver returns before issuing it)
    }
    used = at + bytes;
    return base + at;
  }
  float* floats(size_t n) { return static_cast<float*>(take(n * sizeof(float))); }
};

struct Half {      // one side of the pairs: packed row range [row0, row0 + rows) and its clouds (starts relative to row0)
  int64_t row0, rows;
  int64_t starts[SE3_MAX_BATCH], lengths[SE3_MAX_BATCH];
  int n;
};

#define SE3_TRY(call)          \
  do {                         \
    const int rc_ = (call);    \
    if (rc_ != SE3_OK) return rc_; \
  } while (0)

int dense(const se3_linear_t& L, const float* x, int64_t rows, float* out, int relu, bool with_bias, void* st) {
  return se3_linear_stream(x, rows, L.in_features, L.in_features, L.pieces, with_bias ? L.bias : nullptr, L.out_features, relu, out,
                           L.out_features, st);
}

// AttentionOutput.forward (output_layer.py:7-22): LN(y + squeeze(relu(expand(y)))), rows x C
int ffn(const se3_layer_t& L, const float* y, int64_t rows, int C, float* out, Arena& S, void* st) {
  float* mid = S.floats((size_t)rows * L.expand.out_features);
  float* sq = S.floats((size_t)rows * C);
  if (!S.ok) return SE3_ERR_WORKSPACE;
  SE3_TRY(dense(L.expand, y, rows, mid, 1, true, st));
  SE3_TRY(dense(L.squeeze, mid, rows, sq, 0, false, st));
  return se3_add_layer_norm_fwd(sq, L.squeeze.bias, y, L.ln2_w, L.ln2_b, rows, rows, C, L.ln2_eps, out, st);
}

int copy_rows(float* dst, int64_t dst_pitch_floats, const float* src, int64_t src_pitch_floats, int64_t width_floats, int64_t height, hipStream_t st) {
  if (hipMemcpy2DAsync(dst, (size_t)dst_pitch_floats * 4, src, (size_t)src_pitch_floats * 4, (size_t)width_floats * 4, (size_t)height,
                       hipMemcpyDeviceToDevice, st) != hipSuccess) {
    se3_set_error("transformer_forward: device copy failed");
    return SE3_ERR_LAUNCH;
  }
  return SE3_OK;
}

int zero(float* p, size_t n, hipStream_t st) {
  if (hipMemsetAsync(p, 0, n * sizeof(float), st) != hipSuccess) {
    se3_set_error("transformer_forward: memset failed");
    return SE3_ERR_LAUNCH;
  }
  return SE3_OK;
}

// key-anchor groups of the equivariant cross attention (se3et_amd/ops.py::cross_eq_groups)
int eq_groups(int A, const Half& q, int H, int C, const Half& k, int64_t v_row_stride) {
  if (C / H != 64 || C % H || A > 6 || v_row_stride % 16) return 1;
  for (int i = 0; i < k.n; i++)
    if (k.starts[i] % 16) return 1;
  int64_t wgs = 0;
  for (int i = 0; i < q.n; i++) wgs += (q.lengths[i] + 127) / 128;
  wgs *= (int64_t)H * A;
  static const int forced = getenv("SE3_EQ_GROUPS") ? atoi(getenv("SE3_EQ_GROUPS")) : 0;      // A/B runs: 2 or 3
  if ((forced == 2 || forced == 3) && A % forced == 0) return forced;
  if (A % 3 == 0 && wgs * 3 <= 320) return 3;
  if (A % 2 == 0 && wgs * 2 <= 320) return 2;
  return 1;
}

}  // namespace

extern "C" void se3_transformer_plan_layout(size_t* layout) {
  if (layout == nullptr) return;
  layout[0] = sizeof(se3_linear_t);
  layout[1] = sizeof(se3_layer_t);
  layout[2] = sizeof(se3_transformer_plan_t);
  layout[3] = offsetof(se3_transformer_plan_t, layers);
  layout[4] = offsetof(se3_transformer_plan_t, starts);
  layout[5] = offsetof(se3_transformer_plan_t, emb);
}

extern "C" size_t se3_transformer_workspace_bytes(const se3_transformer_plan_t* plan) {
  if (plan == nullptr) return 0;
  const size_t A = plan->A, C = plan->C, H = plan->H, R = plan->rows;
  size_t widest = 0, logits = 0;
  for (int i = 0; i < plan->num_blocks; i++) {
    const se3_layer_t& L = plan->layers[i];
    if (L.type <= 1 && (size_t)L.stack.out_features > widest) widest = L.stack.out_features;
  }
  if (widest < 3 * C) widest = 3 * C;
  for (int c = 0; c < 2 * plan->num_pairs; c++) logits += A * H * (size_t)plan->lengths[c] * (((size_t)plan->lengths[c] + 31) / 32 * 32);
  const size_t kv = se3_attention_kv_pieces_bytes((int)A, (int64_t)R, (int)C, (int)R);
  const size_t x6 = se3_cross_eq_x6_workspace_bytes((int)A, (int64_t)R, (int64_t)R, (int)C, (int)R);
  const size_t gram = 4 * A * (size_t)plan->num_pairs * C * C;          // two Gram tensors per direction, two directions per block
  // persistent slots (4 x (A, R, C)) + one block's scratch: projection, values, logits, hidden (up to 3 key-anchor groups), products, FFN,
  // both directions of a cross block (no reuse inside a block)
  const size_t floats = 4 * A * R * C + A * R * (widest + 14 * C) + logits + gram + 65536;
  return floats * sizeof(float) + 2 * kv + 2 * x6 + (1u << 20);
}

extern "C" int se3_transformer_forward(const se3_transformer_plan_t* plan, const float* x_in, float* out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(plan && x_in && out && workspace, SE3_ERR_INVALID_ARG, "transformer_forward: null pointer");
  const int A = plan->A, C = plan->C, H = plan->H, B = plan->num_pairs, NB = plan->num_blocks;
  SE3_REQUIRE(A >= 1 && A <= 6 && C % 32 == 0 && H >= 1 && C % H == 0 && B >= 1 && 2 * B <= SE3_MAX_BATCH && NB >= 1 && NB <= SE3_MAX_BLOCKS,
              SE3_ERR_UNSUPPORTED, "transformer_forward: A %d C %d H %d pairs %d blocks %d", A, C, H, B, NB);
  SE3_REQUIRE(workspace_bytes >= se3_transformer_workspace_bytes(plan) && ((uintptr_t)workspace & 255) == 0, SE3_ERR_WORKSPACE,
              "transformer_forward: workspace too small or not 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t R = plan->rows, R0 = plan->rows0, R1 = R - R0;
  Half P0{}, P1{};
  P0.row0 = 0; P0.rows = R0; P0.n = B;
  P1.row0 = R0; P1.rows = R1; P1.n = B;
  for (int i = 0; i < B; i++) {
    P0.starts[i] = plan->starts[i];
    P0.lengths[i] = plan->lengths[i];
    P1.starts[i] = plan->starts[B + i] - R0;
    P1.lengths[i] = plan->lengths[B + i];
    SE3_REQUIRE(plan->starts[i] % 32 == 0 && plan->starts[B + i] % 32 == 0 && plan->starts[B + i] >= R0 && plan->lengths[i] >= 1 &&
                    plan->lengths[B + i] >= 1,
                SE3_ERR_INVALID_ARG, "transformer_forward: cloud %d layout", i);
  }
  Arena W{static_cast<unsigned char*>(workspace), workspace_bytes, 0, true};
  const size_t slot_floats = (size_t)A * R * C;
  float* slot[4];
  for (int i = 0; i < 4; i++) slot[i] = W.floats(slot_floats);
  const size_t scratch0 = W.used;
  // the state of the scheduler: X (equivariant (A, R, C) or invariant (R, C)), and the anchor features kept beside their anchor maximum
  const float* X = x_in;
  bool X_is_eq = true;
  const float* Xeq = nullptr;
  auto free_slot = [&](const float* a, const float* b, const float* c = nullptr) {
    for (int i = 0; i < 4; i++)
      if (slot[i] != a && slot[i] != b && slot[i] != c) return slot[i];
    return slot[0];
  };
  const float scale = 1.0f / std::sqrt((float)(C / H));

  for (int bi = 0; bi < NB; bi++) {
    const se3_layer_t& L = plan->layers[bi];
    const int next = bi + 1 < NB ? plan->layers[bi + 1].type : -1, prev = bi > 0 ? plan->layers[bi - 1].type : -1;
    W.used = scratch0;
    if (L.type <= 1) {
      // ---------------- RPE self attention of all clouds (rpe_transformer.py:174-194), packed rows ----------------
      const float* src = Xeq ? Xeq : X;
      const int As = (Xeq || X_is_eq) ? A : 1;
      const int64_t rows = (int64_t)As * R;
      const int NS = L.stack.out_features;
      const bool eq = L.type == 1 && L.off_qe >= 0;
      float* proj = W.floats((size_t)rows * NS);
      float* vt = W.floats((size_t)As * C * R);
      float* hidden = W.floats((size_t)rows * C);
      size_t lg = 0;
      for (int c = 0; c < 2 * B; c++) lg += (size_t)As * H * plan->lengths[c] * ((plan->lengths[c] + 31) / 32 * 32);
      float* logits = W.floats(lg);
      const size_t kvb = se3_attention_kv_pieces_bytes(As, R, C, (int)R);
      void* kv = kvb ? W.take(kvb) : nullptr;
      float* lin = W.floats((size_t)rows * C);
      float* y = W.floats((size_t)rows * C);
      if (!W.ok) { se3_set_error("transformer_forward: workspace exhausted (self block %d)", bi); return SE3_ERR_WORKSPACE; }
      SE3_TRY(dense(L.stack, src, rows, proj, 0, true, stream));
      SE3_TRY(se3_linear_stream_transposed(src, rows, C, C, L.v.pieces, L.v.bias, C, (int)R, vt, R, stream));
      SE3_TRY(zero(hidden, (size_t)rows * C, st));
      const float* const* eqp = eq ? plan->eq : nullptr;
      const float* qe = eq ? proj + L.off_qe : nullptr;
      if (plan->emb_bf16)
        SE3_TRY(se3_rpe_self_attention_stack_bf16_fwd(proj + L.off_q, proj + L.off_k, vt, proj + L.off_qp, qe, NS, As > 1 ? (int64_t)R * NS : 0, (int)R,
                                                      As > 1 ? (int64_t)C * R : 0, reinterpret_cast<const uint16_t* const*>(plan->emb), eqp, plan->starts,
                                                      plan->lengths, 2 * B, As, C, H, logits, As > 1 ? (int64_t)R * C : 0, hidden, kv, kvb, stream));
      else
        SE3_TRY(se3_rpe_self_attention_stack_fwd(proj + L.off_q, proj + L.off_k, vt, proj + L.off_qp, qe, NS, As > 1 ? (int64_t)R * NS : 0, (int)R,
                                                 As > 1 ? (int64_t)C * R : 0, plan->emb, eqp, plan->starts, plan->lengths, 2 * B, As, C, H, logits,
                                                 As > 1 ? (int64_t)R * C : 0, hidden, kv, kvb, stream));
      SE3_TRY(dense(L.out, hidden, rows, lin, 0, false, stream));
      SE3_TRY(se3_add_layer_norm_fwd(lin, L.out.bias, src, L.ln1_w, L.ln1_b, rows, rows, C, L.ln1_eps, y, stream));
      float* Xn = free_slot(X, Xeq);
      SE3_TRY(ffn(L, y, rows, C, Xn, W, stream));
      if (L.type == 1 && next == 2) {             // anchor features kept for the plain cross block that follows (SE3ET-I)
        float* inv = free_slot(Xn, X, Xeq);
        SE3_TRY(se3_anchor_max(Xn, A, R, C, (int64_t)R * C, C, inv, stream));
        Xeq = Xn;
        X = inv;
        X_is_eq = false;
      } else {
        X = Xn;
        X_is_eq = As > 1;
        // (Xeq keeps its value, as in the scheduler of the reference: only 'cross' blocks and eq2inv replace it)
      }
    } else if (L.type == 2) {
      // ---------------- plain cross attention, ref <- src then src <- updated ref (conditional_transformer.py:304-305,333-334) ----------------
      SE3_REQUIRE(!X_is_eq, SE3_ERR_UNSUPPORTED, "transformer_forward: block %d: plain cross attention on anchor features", bi);
      const bool eq_values = next == 1 || (next < 0 && prev == 1);
      SE3_REQUIRE(!eq_values || Xeq != nullptr, SE3_ERR_INVALID_ARG, "transformer_forward: block %d needs the anchor features of a self_eq block", bi);
      const int Av = eq_values ? A : 1;
      float* Xn = free_slot(X, Xeq);                  // the invariant result (R, C): x0 rows then x1 rows
      float* Xeqn = eq_values ? free_slot(X, Xeq, Xn) : nullptr;
      // one direction: queries xq (Rq, C) of half Pq, keys xk (Rk, C) and values xv ((Av, Rk, C), anchor pitch xv_pitch) of half Pk
      auto direction = [&](const Half& Pq, const Half& Pk, const float* xq, const float* xk, const float* xv, int64_t xv_pitch, float* y_out /* (Av, Rq, C) */) -> int {
        const int64_t Rq = Pq.rows, Rk = Pk.rows;
        float* q = W.floats((size_t)Rq * C);
        float* k = W.floats((size_t)Rk * C);
        float* xvc = W.floats((size_t)Av * Rk * C);
        float* vt = W.floats((size_t)Av * C * Rk);
        float* hidden = W.floats((size_t)Av * Rq * C);
        float* lin = W.floats((size_t)Av * Rq * C);
        float* y = W.floats((size_t)Av * Rq * C);
        if (!W.ok) { se3_set_error("transformer_forward: workspace exhausted (cross block %d)", bi); return SE3_ERR_WORKSPACE; }
        SE3_TRY(dense(L.q, xq, Rq, q, 0, true, stream));
        SE3_TRY(dense(L.k, xk, Rk, k, 0, true, stream));
        const float* xvv = xv;
        if (Av > 1 && xv_pitch != Rk * C) {           // a row range of (A, R, C): made contiguous
          SE3_TRY(copy_rows(xvc, Rk * C, xv, xv_pitch, Rk * C, Av, st));
          xvv = xvc;
        }
        SE3_TRY(se3_linear_stream_transposed(xvv, (int64_t)Av * Rk, C, C, L.v.pieces, L.v.bias, C, (int)Rk, vt, Rk, stream));
        SE3_TRY(zero(hidden, (size_t)Av * Rq * C, st));
        SE3_TRY(se3_attention_stack_fwd(q, k, vt, nullptr, Pq.starts, Pq.lengths, Pk.starts, Pk.lengths, nullptr, Pq.n, Av, C, H, C, C, (int)Rk, 0, 0,
                                        Av > 1 ? (int64_t)C * Rk : 0, Av > 1 ? (int64_t)Rq * C : 0, scale, hidden, nullptr, 0, stream));
        SE3_TRY(dense(L.out, hidden, (int64_t)Av * Rq, lin, 0, false, stream));
        SE3_TRY(se3_add_layer_norm_fwd(lin, L.out.bias, xq, L.ln1_w, L.ln1_b, (int64_t)Av * Rq, Rq, C, L.ln1_eps, y, stream));      // (residual broadcast over the anchors)
        return ffn(L, y, (int64_t)Av * Rq, C, y_out, W, stream);
      };
      if (eq_values) {
        float* y0 = W.floats((size_t)A * R0 * C);
        float* y1 = W.floats((size_t)A * R1 * C);
        SE3_TRY(direction(P0, P1, X, X + R0 * C, Xeq + R0 * C, (int64_t)R * C, y0));
        SE3_TRY(se3_anchor_max(y0, A, R0, C, (int64_t)R0 * C, C, Xn, stream));
        SE3_TRY(direction(P1, P0, X + R0 * C, Xn, y0, (int64_t)R0 * C, y1));
        SE3_TRY(se3_anchor_max(y1, A, R1, C, (int64_t)R1 * C, C, Xn + R0 * C, stream));
        SE3_TRY(copy_rows(Xeqn, R * C, y0, R0 * C, R0 * C, A, st));
        SE3_TRY(copy_rows(Xeqn + R0 * C, R * C, y1, R1 * C, R1 * C, A, st));
        Xeq = Xeqn;
      } else {
        SE3_TRY(direction(P0, P1, X, X + R0 * C, X + R0 * C, 0, Xn));
        SE3_TRY(direction(P1, P0, X + R0 * C, Xn, Xn, 0, Xn + R0 * C));
      }
      X = Xn;
      X_is_eq = false;
    } else {
      // ---------------- anchor-equivariant cross attention (vanilla_transformer.py:751-870), both directions ----------------
      SE3_REQUIRE(X_is_eq && A == 6, SE3_ERR_UNSUPPORTED, "transformer_forward: block %d: equivariant cross attention needs (6, rows, C) features", bi);
      const int mode = L.type == 3 ? 0 : 1;
      float* y0 = W.floats((size_t)A * R0 * C);
      float* y1 = W.floats((size_t)A * R1 * C);
      float* mix0 = W.floats((size_t)B * A * A);
      float* mix1 = W.floats((size_t)B * A * A);
      auto direction = [&](const Half& Pq, const Half& Pk, const float* xq /* (A, Rq, C) contiguous */, const float* xk, float* mix, float* y_out) -> int {
        const int64_t Rq = Pq.rows, Rk = Pk.rows;
        float* q = W.floats((size_t)A * Rq * C);
        float* k = W.floats((size_t)A * Rk * C);
        float* vt = W.floats((size_t)A * C * Rk);
        const int G = eq_groups(A, Pq, H, C, Pk, Rk);
        float* hidden = W.floats((size_t)A * Rq * G * C);
        float* gq = W.floats((size_t)A * B * C * C);
        float* gk = W.floats((size_t)A * B * C * C);
        float* partial = W.floats((size_t)B * A * A);
        float* weights = W.floats((size_t)B * (mode == 0 ? A * A : L.num_rotations));
        const size_t x6b = se3_cross_eq_x6_workspace_bytes(A, Rq, Rk, C, (int)Rk);
        void* x6 = W.take(x6b);
        float* lin = W.floats((size_t)A * Rq * C);
        float* y = W.floats((size_t)A * Rq * C);
        if (!W.ok) { se3_set_error("transformer_forward: workspace exhausted (equivariant cross block %d)", bi); return SE3_ERR_WORKSPACE; }
        SE3_TRY(dense(L.q, xq, (int64_t)A * Rq, q, 0, true, stream));
        SE3_TRY(dense(L.k, xk, (int64_t)A * Rk, k, 0, true, stream));
        SE3_TRY(se3_linear_stream_transposed(xk, (int64_t)A * Rk, C, C, L.v.pieces, L.v.bias, C, (int)Rk, vt, Rk, stream));
        SE3_TRY(zero(hidden, (size_t)A * Rq * G * C, st));
        SE3_TRY(se3_gram_stack(q, A, C, (int64_t)Rq * C, Pq.starts, Pq.lengths, B, gq, stream));
        SE3_TRY(se3_gram_stack(k, A, C, (int64_t)Rk * C, Pk.starts, Pk.lengths, B, gk, stream));
        const float f = 1.0f / (std::sqrt((float)(C / H)) * (float)H);
        SE3_TRY(se3_gram_frobenius(gq, gk, A, B, (int64_t)C * C, f * f, partial, stream));
        SE3_TRY(se3_cross_eq_stack_x6_fwd(q, k, vt, Pq.starts, Pq.lengths, Pk.starts, Pk.lengths, B, A, C, H, Rq, Rk, (int64_t)Rq * C, (int64_t)Rk * C,
                                          (int)Rk, (int64_t)C * Rk, mode, L.trace_idx, L.num_rotations, 1, partial, mix, weights, hidden, G,
                                          (int64_t)Rq * G * C, x6, x6b, stream));
        const void* wp = G == 1 ? L.out.pieces : (G == 2 ? L.out_pieces_g2 : L.out_pieces_g3);
        SE3_REQUIRE(wp != nullptr, SE3_ERR_INVALID_ARG, "transformer_forward: block %d: no output projection for %d key-anchor groups", bi, G);
        SE3_TRY(se3_linear_stream(hidden, (int64_t)A * Rq, G * C, (int64_t)G * C, wp, nullptr, C, 0, lin, C, stream));
        SE3_TRY(se3_add_layer_norm_fwd(lin, L.out.bias, xq, L.ln1_w, L.ln1_b, (int64_t)A * Rq, (int64_t)A * Rq, C, L.ln1_eps, y, stream));
        return ffn(L, y, (int64_t)A * Rq, C, y_out, W, stream);
      };
      // the halves of X (A, R, C) as contiguous (A, R0, C) / (A, R1, C)
      float* xq0 = W.floats((size_t)A * R0 * C);
      float* xk0 = W.floats((size_t)A * R1 * C);
      if (!W.ok) { se3_set_error("transformer_forward: workspace exhausted (equivariant cross block %d)", bi); return SE3_ERR_WORKSPACE; }
      SE3_TRY(copy_rows(xq0, R0 * C, X, R * C, R0 * C, A, st));
      SE3_TRY(copy_rows(xk0, R1 * C, X + R0 * C, R * C, R1 * C, A, st));
      SE3_TRY(direction(P0, P1, xq0, xk0, mix0, y0));
      SE3_TRY(direction(P1, P0, xk0, y0, mix1, y1));
      float* Xn = free_slot(X, Xeq);
      if (L.type == 4 && next >= 0 && next != 1 && next < 3) {
        // eq2inv_soft (conditional_transformer.py:209-249): the src features in the frame the ref <- src rotation weights prefer, then both
        // sides compressed over the anchors (output_layer.py:24-47)
        SE3_REQUIRE(plan->rc_expand.pieces && plan->rc_squeeze.pieces, SE3_ERR_INVALID_ARG, "transformer_forward: block %d needs the rotcompress layer", bi);
        float* y1p = W.floats((size_t)A * R1 * C);
        SE3_TRY(se3_anchor_mix_stack(y1, R1, C, mix0, P1.starts, P1.lengths, B, y1p, stream));
        auto rotcompress = [&](const float* x, int64_t Rh, float* o) -> int {
          float* mx = W.floats((size_t)Rh * C);
          float* mid = W.floats((size_t)Rh * plan->rc_expand.out_features);
          float* sq = W.floats((size_t)Rh * C);
          if (!W.ok) { se3_set_error("transformer_forward: workspace exhausted (rotcompress)"); return SE3_ERR_WORKSPACE; }
          SE3_TRY(se3_anchor_max(x, A, Rh, C, Rh * C, C, mx, stream));
          SE3_TRY(se3_linear_stream_segments(x, Rh, A * C, C, Rh * C, plan->rc_expand.pieces, plan->rc_expand.bias, plan->rc_expand.out_features, 1, mid,
                                             plan->rc_expand.out_features, stream));
          SE3_TRY(dense(plan->rc_squeeze, mid, Rh, sq, 0, false, stream));
          return se3_add_layer_norm_fwd(sq, plan->rc_squeeze.bias, mx, plan->rc_ln_w, plan->rc_ln_b, Rh, Rh, C, plan->rc_ln_eps, o, stream);
        };
        SE3_TRY(rotcompress(y0, R0, Xn));
        SE3_TRY(rotcompress(y1p, R1, Xn + R0 * C));
        X = Xn;
        X_is_eq = false;
        Xeq = nullptr;
      } else {
        SE3_TRY(copy_rows(Xn, R * C, y0, R0 * C, R0 * C, A, st));
        SE3_TRY(copy_rows(Xn + R0 * C, R * C, y1, R1 * C, R1 * C, A, st));
        X = Xn;
        X_is_eq = true;
      }
    }
  }
  SE3_REQUIRE(!X_is_eq, SE3_ERR_UNSUPPORTED, "transformer_forward: the block list must end on invariant features");
  return dense(plan->out_proj, X, R, out, 0, true, stream);
}
