"""Autograd for the HIP ops (BASELINE.json configs[4]).

Forward = the gfx950 kernel.  Backward: hand-written for the ops that dominated the step (KPConv: csrc/kpconv_so3.hip kpconv_scatter_kernel
+ two library GEMMs, 474 -> 5 ms per 5k+5k step; GroupNorm: csrc/rowops.hip gn_bwd_*, 20 -> 5 ms; Sinkhorn: csrc/sinkhorn.hip sinkhorn_bwd_kernel, 18 -> 0.4 ms; geometric embedding: GEMM operands from
csrc/geo_embedding.hip, 13 -> 3 ms; neighbour max-pool; padded row gather; the three attention ops: se3et_amd/attention_bwd.py,
logits recomputed by the forward's kernel + batched GEMMs, 12.9 -> 8.1 ms --
`hip_backward`), and for anything else (and as the pin of the hand-derived forms: functional.ATTENTION_BACKWARD = 'autograd') reverse-mode
differentiation of a PyTorch restatement of the SAME op, re-evaluated on the GPU inside backward ('until a
backward kernel exists autograd runs through the PyTorch restatement' -- `differentiable`).  Every
restatement below is plain torch on GPU tensors -- nothing here runs on the CPU and nothing imports oracle/.  `differentiable(hip_fn,
torch_fn, *tensors)` is the single mechanism: it calls the kernel under no_grad, keeps the inputs, and in backward builds the torch
graph of `torch_fn` on detached copies and pulls the incoming gradients through it.  The restatements are pinned twice: forward
values against the kernels (tests/test_gpu_training.py::test_restatements_match_the_kernels) and losses / gradients against the
genuine reference's training step (tests/golden/train_micro_*.npz).

Reference formulas: blocks_epn.py:334-546 (KPConvInterSO3), :684-701 (GroupNormEPN), blocks.py:93-110 (max_pool),
geotransformer.py:57-121 (embeddings), rpe_transformer.py:39-131, vanilla_transformer.py:39-85, :247-641, learnable_sinkhorn.py:13-66."""
import math

import torch
import torch.nn.functional as F


def _op_name(fn):
    name = getattr(fn, '__name__', 'op')
    if name == '<lambda>':       # the restatement the lambda forwards to
        name = next((n for n in fn.__code__.co_names if n in globals() and callable(globals()[n])), name)
    return name


BACKWARD_TIMINGS = None      # dict op name -> list of (start_event, end_event) while enabled (tools/train_bench.py --profile)


class _HipForwardTorchBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hip_fn, torch_fn, num_outputs, *tensors):
        ctx.torch_fn = torch_fn
        ctx.save_for_backward(*[t for t in tensors if t is not None])
        ctx.present = [t is not None for t in tensors]
        with torch.no_grad():
            out = hip_fn(*tensors)
        if num_outputs == 1:
            return out
        ctx.mark_non_differentiable(*[o for o in out if not o.is_floating_point()])
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        if BACKWARD_TIMINGS is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                return _HipForwardTorchBackward._backward(ctx, *grads)
            finally:
                e1.record()
                BACKWARD_TIMINGS.setdefault(_op_name(ctx.torch_fn), []).append((e0, e1))
        return _HipForwardTorchBackward._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, *grads):
        saved = list(ctx.saved_tensors)
        inputs, it = [], iter(saved)
        for present in ctx.present:
            inputs.append(next(it) if present else None)
        leaves = [t.detach().requires_grad_(True) if (t is not None and t.is_floating_point() and need) else (t.detach() if t is not None else None)
                  for t, need in zip(inputs, ctx.needs_input_grad[3:])]
        with torch.enable_grad():
            out = ctx.torch_fn(*leaves)
        outs = [out] if torch.is_tensor(out) else list(out)
        pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
        wanted = [t for t in leaves if t is not None and t.requires_grad]
        got = torch.autograd.grad([o for o, _ in pairs], wanted, [g for _, g in pairs], allow_unused=True) if pairs and wanted else ()
        gi = iter(got)
        result = [None, None, None]
        for t in leaves:
            result.append(next(gi) if (t is not None and t.requires_grad) else None)
        return tuple(result)


PROFILE_RANGES = False


class _HipForwardHipBackward(torch.autograd.Function):
    """Forward and backward both hand-written: bwd_fn(grad_out, needs, *tensors) -> one gradient (or None) per tensor."""

    @staticmethod
    def forward(ctx, hip_fn, bwd_fn, name, *tensors):
        ctx.bwd_fn, ctx.name = bwd_fn, name
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            out = hip_fn(*tensors)
        if torch.is_tensor(out):
            return out
        ctx.mark_non_differentiable(*out[1:])          # further outputs are constants for autograd (e.g. the equivariant embedding)
        return tuple(out)

    @staticmethod
    def backward(ctx, grad, *unused):
        needs = ctx.needs_input_grad[3:]
        if BACKWARD_TIMINGS is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        with torch.no_grad():
            if PROFILE_RANGES:                          # (tools/r5/train_launches.py: the node's launches under the op's name)
                with torch.autograd.profiler.record_function('hipbwd:' + ctx.name):
                    got = ctx.bwd_fn(grad, needs, *ctx.saved_tensors)
            else:
                got = ctx.bwd_fn(grad, needs, *ctx.saved_tensors)
        if BACKWARD_TIMINGS is not None:
            e1.record()
            BACKWARD_TIMINGS.setdefault(ctx.name + ' (HIP)', []).append((e0, e1))
        return (None, None, None) + tuple(g if n else None for g, n in zip(got, needs))


def hip_backward(hip_fn, bwd_fn, name, *tensors):
    """hip_fn(*tensors) (one differentiable output, the first) with the hand-written backward bwd_fn(grad_out, needs_input_grad, *tensors)."""
    return _HipForwardHipBackward.apply(hip_fn, bwd_fn, name, *tensors)


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def differentiable(hip_fn, torch_fn, num_outputs, *tensors):
    """hip_fn(*tensors) with a backward through torch_fn(*tensors); None entries are passed through as None."""
    return _HipForwardTorchBackward.apply(hip_fn, torch_fn, num_outputs, *tensors)


# ---------------------------------------------------------------------------------------------------------------------------------------
# PyTorch restatements (GPU tensors, differentiable)
# ---------------------------------------------------------------------------------------------------------------------------------------
def _padded_rows(x, idx):
    """x[idx] where idx == x.shape[0] (the padding index), or a negative width marker, addresses an all-zero row."""
    n = x.shape[0]
    xs = torch.cat((x, x.new_zeros((1,) + tuple(x.shape[1:]))), 0)
    return xs[torch.where((idx < 0) | (idx > n), torch.full_like(idx, n), idx)]


def kpconv_inter_so3(x, q_pts, s_pts, idx, kernel_points, weights, kidx, ridx, sigma):
    """blocks_epn.py:454-546: out[p, r, d] = sum_{k, a, c} F[p, k, a, c] W[kidx[k, r], ridx[a, r], c, d]."""
    ns = s_pts.shape[0]
    safe = torch.where((idx < 0) | (idx > ns), torch.full_like(idx, ns), idx)
    sp = torch.cat((s_pts, torch.full_like(s_pts[:1], 1e6)), 0)
    nb = sp[safe] - q_pts[:, None]                                                             # (P, NN, 3)
    w = (1.0 - (nb[:, :, None] - kernel_points[None, None]).norm(dim=-1) / sigma).clamp(min=0.0)   # (P, NN, K)
    feats = torch.einsum('pnk,pnac->pkac', w, _padded_rows(x, safe))
    wfull = weights[kidx[:, None, :], ridx[None, :, :]]                                        # (K, A, R, Cin, Cout)
    return torch.einsum('pkac,karcd->prd', feats, wfull)


def group_norm_rows(x, weight, bias, residual, x_bias, groups, eps, leaky_slope, segments):
    """GroupNorm with statistics over (all leading dims x channels of the group) (blocks_epn.py:684-701), optional bias of the
    producing linear layer, residual and LeakyReLU; `segments`: row offsets of independently normalised ranges."""
    C = x.shape[-1]
    y = x.reshape(-1, C)
    if x_bias is not None:
        y = y + x_bias
    bounds = segments if segments is not None else [0, y.shape[0]]
    outs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        g = y[a:b].reshape(b - a, groups, C // groups)
        mean = g.mean((0, 2), keepdim=True)
        var = g.var((0, 2), unbiased=False, keepdim=True)
        outs.append(((g - mean) * torch.rsqrt(var + eps)).reshape(b - a, C))
    y = (torch.cat(outs, 0) if len(outs) > 1 else outs[0]) * weight + bias
    y = y.reshape(x.shape)
    if residual is not None:
        y = y + residual
    return F.leaky_relu(y, leaky_slope) if leaky_slope is not None else y


def neighbor_max_pool(x, idx):
    """blocks.py:93-110: max over the gathered neighbour rows, the zero row included for padded entries."""
    rows = _padded_rows(x, idx)
    if bool((idx < 0).any()):                      # width markers of stacked pairs are skipped, not zero rows
        rows = rows.masked_fill((idx < 0).reshape(idx.shape + (1,) * (rows.dim() - 2)), float('-inf'))
    return rows.amax(1)


def gather_rows_padded(x, idx):
    return _padded_rows(x, idx)


def add_layer_norm(hidden, residual, weight, bias, hidden_bias, eps):
    h = hidden if hidden_bias is None else hidden + hidden_bias
    return F.layer_norm(h + residual, (hidden.shape[-1],), weight, bias, eps)


def sinusoidal_embedding(x, div_term):
    om = x.unsqueeze(-1) * div_term
    return torch.stack((torch.sin(om), torch.cos(om)), -1).flatten(-2)


def geometric_embedding(points, div_term, w_d, b_d, w_a, b_a, knn, sigma_d, sigma_a):
    """geotransformer.py:69-121 with the 3 nearest other points given (`knn`, not differentiable): E (N, N, C)."""
    sq = (points * points).sum(-1)
    d = torch.sqrt((sq[:, None] - 2.0 * points @ points.t() + sq[None, :]).clamp(min=0.0)) / sigma_d
    ref = points[knn] - points[:, None]                                    # (N, 3, 3)
    anc = points[None, :, :] - points[:, None, :]                          # (N, N, 3)
    ref = ref[:, None].expand(-1, points.shape[0], -1, -1)
    anc = anc[:, :, None].expand(-1, -1, knn.shape[1], -1)
    sin = torch.linalg.norm(torch.cross(ref, anc, dim=-1), dim=-1)
    cos = (ref * anc).sum(-1)
    a = torch.atan2(sin, cos) * (180.0 / (sigma_a * math.pi))             # (N, N, 3)
    ed = F.linear(sinusoidal_embedding(d, div_term), w_d, b_d)
    ea = F.linear(sinusoidal_embedding(a, div_term), w_a, b_a).amax(2)
    return ed + ea


def _heads(x, h):
    return x.reshape(x.shape[:-1] + (h, x.shape[-1] // h))


def rpe_attention(q, k, vt, emb, w_p, eq_emb, w_eq, num_heads):
    """rpe_transformer.py:85-131 on projected q ([A,] N, C), k ([A,] M, C), transposed values vt ([A,] C, >= M); the position terms in
    the folded form q.(W e) = (W^T q).e (the bias terms are constant along the softmax axis and cancel)."""
    anchored = q.dim() == 3
    q3, k3, v3 = (q, k, vt) if anchored else (q[None], k[None], vt[None])
    A, N, C = q3.shape
    M, H = k3.shape[1], num_heads
    d = C // H
    qh, kh = _heads(q3, H), _heads(k3, H)                                                     # (A, N, H, d)
    vh = _heads(v3[..., :M].transpose(1, 2), H)                                               # (A, M, H, d)
    scores = torch.einsum('anhd,amhd->ahnm', qh, kh)
    qp = torch.einsum('anhd,hdc->anhc', qh, w_p.reshape(H, d, C))
    scores = scores + torch.einsum('anhc,nmc->ahnm', qp, emb.to(qp.dtype))
    if eq_emb is not None:
        qe = torch.einsum('anhd,hde->anhe', qh, w_eq.reshape(H, d, eq_emb.shape[-1]))
        scores = scores + torch.einsum('anhe,anme->ahnm', qe, eq_emb)
    p = torch.softmax(scores / math.sqrt(d), -1)
    out = torch.einsum('ahnm,amhd->anhd', p, vh).reshape(A, N, C)
    return out if anchored else out[0]


def cross_attention(q, k, vt, num_heads):
    """vanilla_transformer.py:39-85: q (N, C), k (M, C); vt (C, >= M) or per-anchor values (A, C, >= M) sharing the scores."""
    N, C = q.shape
    M, H = k.shape[0], num_heads
    d = C // H
    p = torch.softmax(torch.einsum('nhd,mhd->hnm', _heads(q, H), _heads(k, H)) / math.sqrt(d), -1)
    if vt.dim() == 2:
        return torch.einsum('hnm,mhd->nhd', p, _heads(vt[:, :M].t(), H)).reshape(N, C)
    vh = _heads(vt[..., :M].transpose(1, 2), H)
    return torch.einsum('hnm,amhd->anhd', p, vh).reshape(vt.shape[0], N, C)


def cross_attention_eq(q, k, vt, trace_idx, num_heads, mode):
    """vanilla_transformer.py:247-641, 751-870 for attn_mode 'a_soft' / 'r_soft' ('sq' global weights, mean pooling): returns
    (hidden (A, N, C), weights, mix (A, A)) as se3et_amd.ops.cross_attention_eq."""
    A, N, C = q.shape
    M, H = k.shape[1], num_heads
    d = C // H
    S = torch.einsum('anhd,emhd->aehnm', _heads(q, H), _heads(k, H)) / math.sqrt(d)            # (A, A, H, N, M)
    g = (S.mean(2) ** 2).mean((-1, -2))                                                       # (A, A)
    if mode == 'a_soft':
        w = g / g.sum(1, keepdim=True)
        mix = w
    else:
        R = trace_idx.shape[0]
        wr = g[torch.arange(A, device=q.device)[None, :], trace_idx].mean(1)                   # (R,)
        w = wr / wr.sum()
        mix = torch.zeros((A, A), dtype=q.dtype, device=q.device).index_put_(
            (torch.arange(A, device=q.device)[None, :].expand(R, A).reshape(-1), trace_idx.reshape(-1)),
            w[:, None].expand(R, A).reshape(-1), accumulate=True)
    vh = _heads(vt[..., :M].transpose(1, 2), H)                                                # (A, M, H, d)
    out = torch.einsum('ae,aehnm,emhd->anhd', mix, torch.softmax(S, -1), vh).reshape(A, N, C)
    return out, w, mix


def log_optimal_transport(scores, alpha, row_masks, col_masks, num_iterations, inf):
    """learnable_sinkhorn.py:13-66."""
    B, R, C = scores.shape
    prm = torch.zeros((B, R + 1), dtype=torch.bool, device=scores.device)
    prm[:, :R] = ~row_masks
    pcm = torch.zeros((B, C + 1), dtype=torch.bool, device=scores.device)
    pcm[:, :C] = ~col_masks
    al = alpha.reshape(1, 1, 1)
    z = torch.cat((torch.cat((scores, al.expand(B, R, 1)), -1), al.expand(B, 1, C + 1)), 1)
    z = z.masked_fill(prm[:, :, None] | pcm[:, None, :], -inf)
    nvr, nvc = row_masks.float().sum(1), col_masks.float().sum(1)
    norm = -torch.log(nvr + nvc)
    log_mu = norm[:, None].repeat(1, R + 1)
    log_mu[:, R] = torch.log(nvc) + norm
    log_mu = log_mu.masked_fill(prm, -inf)
    log_nu = norm[:, None].repeat(1, C + 1)
    log_nu[:, C] = torch.log(nvr) + norm
    log_nu = log_nu.masked_fill(pcm, -inf)
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(num_iterations):
        u = log_mu - torch.logsumexp(z + v[:, None, :], 2)
        v = log_nu - torch.logsumexp(z + u[:, :, None], 1)
    return z + u[:, :, None] + v[:, None, :] - norm[:, None, None]
