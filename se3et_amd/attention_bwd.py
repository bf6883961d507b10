"""Hand-derived backward passes of the attention ops (training step).

The forward kernels never keep the (A, H, N, M) probabilities; the backward recomputes them -- the relative-position logits with the same
HIP kernel as the forward (csrc/attention.hip rpe_bias_kernel, the folded form q.(W e) = (W^T q).e of rpe_transformer.py:85-131) -- and
contracts on the library's batched GEMMs.  The (N, M, C) embedding is read twice (logits, d folded query) and its gradient written once:
autograd through the PyTorch restatement (se3et_amd/autograd.py) moved it seven times (einsum operand copies and the permuted gradient).
No probabilities, no (N, M, H, C) position tensors, no autograd graph.

    S = (q k^T + qp emb^T [+ qe eq_emb^T]) / sqrt(d),  P = softmax_m S,  O = P v
    dV = P^T dO,  dP = dO v^T,  dS = P o (dP - rowsum(dP o P)) / sqrt(d)
    dq = dS k [+ dqp W_p^T + dqe W_eq^T],  dk = dS^T q,  dqp[a, n, h] = sum_m dS[a, h, n, m] emb[n, m],
    demb[n, m] = sum_{a, h} dS[a, h, n, m] qp[a, n, h],  dW_p = sum q^T dqp

Pinned against reverse-mode differentiation of the restatements: tests/test_gpu_training.py::test_attention_backward_matches_autograd, and
against the genuine reference's training step through the C5 fixtures.
"""
import math

import torch

from . import ops as _ops


def _heads_first(x, H):
    """(A, R, H * d) -> (A * H, R, d) contiguous."""
    A, R, C = x.shape
    return x.reshape(A, R, H, C // H).permute(0, 2, 1, 3).reshape(A * H, R, C // H)


def _heads_last(x, A, H):
    """(A * H, R, d) -> (A, R, H * d)."""
    AH, R, d = x.shape
    return x.reshape(A, H, R, d).permute(0, 2, 1, 3).reshape(A, R, H * d)


def rpe_attention_bwd(grad, needs, q, k, vt, emb, w_p, eq_emb, w_eq, num_heads):
    """Gradients of functional.rpe_attention's hidden states w.r.t. (q, k, vt, emb, w_p, eq_emb (none), w_eq)."""
    anchored = q.dim() == 3
    q3, k3, v3, g3 = (q, k, vt, grad) if anchored else (q[None], k[None], vt[None], grad[None])
    A, N, C = q3.shape
    M, H = emb.shape[1], num_heads
    d = C // H
    scale = 1.0 / math.sqrt(d)
    emb = emb.float()
    qh = _heads_first(q3, H)                                              # (AH, N, d)
    kh = _heads_first(k3, H)                                              # (AH, M, d)
    vh = v3[..., :M].reshape(A * H, d, M)                                 # (AH, d, M): the transposed values are already head-major
    gh = _heads_first(g3.contiguous(), H)                                 # (AH, N, d)
    wp = w_p.reshape(H, d, C)
    # folded queries and logits as the forward computes them
    qp = torch.matmul(qh.reshape(A, H, N, d), wp)                         # (A, H, N, C)
    qp_rows = qp.permute(0, 2, 1, 3).reshape(A, N, H * C)
    qe_rows = None
    if eq_emb is not None:
        we = w_eq.reshape(H, d, eq_emb.shape[-1])
        qe = torch.matmul(qh.reshape(A, H, N, d), we)                     # (A, H, N, 4)
        both = torch.cat((qp_rows, qe.permute(0, 2, 1, 3).reshape(A, N, -1)), -1)
        qp_rows, qe_rows = both[..., :H * C], both[..., H * C:]
    bias = _ops.rpe_bias(qp_rows, qe_rows, emb, eq_emb, H)                # (AH, N, Mp)
    S = torch.baddbmm(bias[..., :M], qh, kh.transpose(1, 2))
    P = torch.softmax(S * scale, -1)                                      # (AH, N, M)
    dV = torch.bmm(gh.transpose(1, 2), P)                                 # (AH, d, M)
    dP = torch.bmm(gh, vh)                                                # (AH, N, M)
    dS = P * (dP - (dP * P).sum(-1, keepdim=True)) * scale
    dq = torch.bmm(dS, kh)                                                # (AH, N, d)
    dk = torch.bmm(dS.transpose(1, 2), qh)                                # (AH, M, d)
    # position terms: one batched GEMM per direction with the query row as the batch
    dSn = dS.permute(1, 0, 2).contiguous()                                # (N, AH, M)
    dqp = torch.bmm(dSn, emb)                                             # (N, AH, C)
    demb = None
    if needs[3]:
        demb = torch.bmm(dSn.transpose(1, 2), qp.reshape(A * H, N, C).permute(1, 0, 2))        # (N, M, C)
    dqp = dqp.permute(1, 0, 2).reshape(A, H, N, C)
    dq = dq + torch.matmul(dqp, wp.transpose(1, 2)).reshape(A * H, N, d)
    dw_p = torch.matmul(qh.reshape(A, H, N, d).transpose(2, 3), dqp).sum(0).reshape(w_p.shape) if needs[4] else None
    dw_eq = None
    if eq_emb is not None:
        # dqe[a, n, h, e] = sum_m dS[a, h, n, m] eq_emb[a, n, m, e]
        dqe = torch.matmul(dS.reshape(A, H, N, M).permute(0, 2, 1, 3), eq_emb)                   # (A, N, H, 4)
        dqe = dqe.permute(0, 2, 1, 3)                                                          # (A, H, N, 4)
        dq = dq + torch.matmul(dqe, we.transpose(1, 2)).reshape(A * H, N, d)
        if needs[6]:
            dw_eq = torch.matmul(qh.reshape(A, H, N, d).transpose(2, 3), dqe).sum(0).reshape(w_eq.shape)
    dq3, dk3 = _heads_last(dq, A, H), _heads_last(dk, A, H)
    dv3 = None
    if needs[2]:
        dv3 = torch.zeros_like(v3)
        dv3[..., :M] = dV.reshape(A, C, M)
    if not anchored:
        dq3, dk3 = dq3[0], dk3[0]
        dv3 = dv3[0] if dv3 is not None else None
    return dq3, dk3, dv3, (demb.to(emb.dtype) if demb is not None else None), dw_p, None, dw_eq


def cross_attention_bwd(grad, needs, q, k, vt, num_heads):
    """Gradients of functional.cross_attention (vanilla_transformer.py:39-85): q (N, C), k (M, C); vt (C, Mp) or per-anchor values
    (A, C, Mp) sharing the scores."""
    N, C = q.shape
    M, H = k.shape[0], num_heads
    d = C // H
    scale = 1.0 / math.sqrt(d)
    qh, kh = _heads_first(q[None], H), _heads_first(k[None], H)           # (H, N, d), (H, M, d)
    P = torch.softmax(torch.bmm(qh, kh.transpose(1, 2)) * scale, -1)      # (H, N, M)
    per_anchor = vt.dim() == 3
    v3 = vt if per_anchor else vt[None]
    g3 = (grad if per_anchor else grad[None]).contiguous()
    A = v3.shape[0]
    vh = v3[..., :M].reshape(A, H, d, M)
    gh = g3.reshape(A, N, H, d).permute(0, 2, 1, 3)                       # (A, H, N, d)
    dP = torch.matmul(gh, vh).sum(0)                                      # (H, N, M)
    dS = P * (dP - (dP * P).sum(-1, keepdim=True)) * scale
    dq = _heads_last(torch.bmm(dS, kh), 1, H)[0]
    dk = _heads_last(torch.bmm(dS.transpose(1, 2), qh), 1, H)[0]
    dvt = None
    if needs[2]:
        dvt = torch.zeros_like(v3)
        dvt[..., :M] = torch.matmul(gh.transpose(2, 3), P).reshape(A, C, M)
        dvt = dvt if per_anchor else dvt[0]
    return dq, dk, dvt


class _CrossAttentionEq(torch.autograd.Function):
    """functional.cross_attention_eq (vanilla_transformer.py:247-641, 751-870; 'a_soft' / 'r_soft') with all three outputs differentiable:
    hidden (A, N, C), the weights ((A, A) or (R,)) and the anchor-pair mixing matrix (eq2inv_soft consumes it).

        S[a,e] = q_a k_e^T / sqrt(d),  g[a,e] = mean_{n,m} (mean_h S)^2,  mix = normalised g (per row / over rotations),
        out[a] = sum_e mix[a,e] softmax_m(S[a,e]) v_e
        dS = mix o P o (G - rowsum(P o G)) + 2 dg mean_h S / (N M H),   G[a,e] = dO_a v_e^T,   dmix[a,e] = <P[a,e], G[a,e]> + g_mix
    """

    @staticmethod
    def forward(ctx, q, k, vt, trace_idx, num_heads, mode):
        with torch.no_grad():
            out, w, mix = _ops.cross_attention_eq(q, k, vt, num_heads, mode, trace_idx)
        ctx.save_for_backward(q, k, vt, trace_idx)
        ctx.num_heads, ctx.mode = num_heads, mode
        return out, w, mix

    @staticmethod
    def backward(ctx, g_out, g_w, g_mix):
        from . import autograd as AG
        if AG.BACKWARD_TIMINGS is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        with torch.no_grad():
            got = _cross_attention_eq_bwd(ctx, g_out, g_w, g_mix)
        if AG.BACKWARD_TIMINGS is not None:
            e1.record()
            AG.BACKWARD_TIMINGS.setdefault('cross_attention_eq (HIP)', []).append((e0, e1))
        return got + (None, None, None)


def _cross_attention_eq_bwd(ctx, g_out, g_w, g_mix):
    q, k, vt, trace_idx = ctx.saved_tensors
    H, mode = ctx.num_heads, ctx.mode
    A, N, C = q.shape
    M = k.shape[1]
    d = C // H
    scale = 1.0 / math.sqrt(d)
    qh = q.reshape(A, N, H, d).permute(0, 2, 1, 3)                        # (A, H, N, d)
    kh = k.reshape(A, M, H, d).permute(0, 2, 1, 3)                        # (A, H, M, d)
    vhT = vt[..., :M].reshape(A, H, d, M)                                 # (A, H, d, M)
    S = torch.matmul(qh[:, None], kh[None].transpose(-1, -2)) * scale     # (A, A, H, N, M)
    P = torch.softmax(S, -1)
    Sbar = S.mean(2)                                                      # (A, A, N, M)
    g = (Sbar * Sbar).mean((-1, -2))                                      # (A, A)
    ar = torch.arange(A, device=q.device)
    if mode == 'a_soft':
        Gs = g.sum(1, keepdim=True)
        mix = g / Gs
    else:
        R = trace_idx.shape[0]
        wr = g[ar[None, :], trace_idx].mean(1)                            # (R,)
        W = wr.sum()
        w = wr / W
        mix = torch.zeros((A, A), dtype=q.dtype, device=q.device).index_put_(
            (ar[None, :].expand(R, A).reshape(-1), trace_idx.reshape(-1)), w[:, None].expand(R, A).reshape(-1), accumulate=True)
    dmix = torch.zeros_like(mix) if g_mix is None else g_mix.reshape(A, A).clone()
    dS = None
    dq = torch.zeros_like(qh)
    dk = torch.zeros_like(kh)
    dvt = torch.zeros_like(vt) if ctx.needs_input_grad[2] else None
    if g_out is not None:
        gh = g_out.reshape(A, N, H, d).permute(0, 2, 1, 3)                # (A, H, N, d)
        G = torch.matmul(gh[:, None], vhT[None])                          # (A, A, H, N, M): dO_a v_e^T
        PG = P * G
        dmix = dmix + PG.sum((-1, -2, -3))
        dS = (PG - P * PG.sum(-1, keepdim=True)) * mix[:, :, None, None, None]
        if dvt is not None:
            # dv_e = sum_a mix[a, e] P[a, e]^T dO_a, stored transposed (A, C, Mp)
            Pw = P * mix[:, :, None, None, None]
            dvt[..., :M] = torch.matmul(gh[:, None].transpose(-1, -2), Pw).sum(0).reshape(A, C, M)
    # through the normalised anchor-pair statistics
    if mode == 'a_soft':
        dw = dmix if g_w is None else dmix + g_w.reshape(A, A)
        dg = dw / Gs - (dw * g).sum(1, keepdim=True) / (Gs * Gs)
    else:
        dw = dmix[ar[None, :], trace_idx].sum(1)                          # (R,)
        if g_w is not None:
            dw = dw + g_w.reshape(-1)
        dwr = dw / W - (dw * wr).sum() / (W * W)
        dg = torch.zeros_like(g).index_put_((ar[None, :].expand_as(trace_idx).reshape(-1), trace_idx.reshape(-1)),
                                            (dwr[:, None] / A).expand_as(trace_idx).reshape(-1), accumulate=True)
    dSg = (dg[:, :, None, None] * Sbar) * (2.0 / (N * M * H))             # (A, A, N, M), the same for every head
    dS = dSg[:, :, None] + dS if dS is not None else dSg[:, :, None].expand(A, A, H, N, M)
    dq = torch.matmul(dS, kh[None]).sum(1) * scale                        # (A, H, N, d)
    dk = torch.matmul(dS.transpose(-1, -2), qh[:, None]).sum(0) * scale   # (A, H, M, d)
    return (dq.permute(0, 2, 1, 3).reshape(A, N, C), dk.permute(0, 2, 1, 3).reshape(A, M, C), dvt)


def cross_attention_eq(q, k, vt, num_heads, mode, trace_idx):
    return _CrossAttentionEq.apply(q, k, vt, trace_idx, num_heads, mode)
