"""Functional layer between the module mirror (se3et_amd.modules) and the C ABI (se3et_amd.ops).

Every op of the SE3ET hot path enters here.  Ops listed in HIP_OPS run as hand-written gfx950
kernels from libse3et_hip.so; dense Linear layers are plain library GEMMs (rocBLAS/hipBLASLt through torch, as the
design allows for non-fused GEMMs).  Nothing here runs on the CPU: tensors must live on the GPU and a missing
library raises (se3et_amd._lib).
"""
import math
import os

import threading

import torch
import torch.nn.functional as F

from . import attention_bwd as ABW
from . import autograd as AG
from . import ops as _ops

ATTENTION_BACKWARD = 'explicit'      # 'autograd': reverse-mode differentiation of the PyTorch restatements (the pin of the explicit form)
HIP_OPS = set()          # filled below as kernels are bound; tests assert the hot ops are in here


def _hip(fn):
    HIP_OPS.add(fn.__name__)
    return fn


# ---------------------------------------------------------------------------------------------------------------------
# dense layers (library GEMMs) and small glue
# ---------------------------------------------------------------------------------------------------------------------
mm = _ops.mm
shared_tensors = _ops._Shared          # device tensors cached across host threads / streams (see ops._Shared)
to_device = _ops.to_device


LINEAR_STREAM = os.environ.get('SE3_LINEAR_STREAM', '1') != '0'          # False: round-3 routing (tile kernel for the long-K shapes, library GEMMs otherwise: A/B runs)


def linear(x, weight, bias=None, relu=False):
    """x W^T [+ b] [ReLU]: the f16 hi / lo split kernel (csrc/linear_f16.hip, f32 accuracy) for the large inference GEMMs, a library GEMM
    otherwise (with relu=True the activation rides in the GEMM epilogue either way)."""
    if AG.needs_grad(x, weight, bias):                      # training: the library GEMM through torch's own autograd
        y = F.linear(x, weight, bias)
        return F.relu(y) if relu else y
    if LINEAR_STREAM:
        # round 4: every inference dense layer on the streaming f16-split kernel (csrc/dense_norm.hip plain mode; faster than the library
        # and than the tile kernel on every transformer shape, tools/micro/linear_stream_shapes.py -> profiles/r04_linear_stream_shapes.txt)
        xs = x if (x.is_contiguous() or x.dim() == 2) else x.contiguous()       # (a row range of packed (A, R, C) features: copied)
        if _ops.linear_stream_ok(xs, weight):
            return _ops.linear_stream(xs, weight, bias, relu)
    if _ops.linear_f16_ok(x, weight):
        return _ops.linear_f16(x, weight, bias, relu)
    if x.dim() == 3 and not x.is_contiguous() and x.stride(2) == 1 and not relu:
        # a row range of packed (A, R, C) features: one batched GEMM over the anchors on the strided view (F.linear would copy it first)
        A, R, N = x.shape[0], x.shape[1], weight.shape[0]
        wt = weight.t()[None].expand(A, weight.shape[1], N)
        return torch.bmm(x, wt) if bias is None else torch.baddbmm(bias[None, None, :].expand(A, R, N), x, wt)
    if bias is None and not relu:
        if x.dim() == 2:
            return mm(x, weight.t())
        return mm(x.reshape(-1, x.shape[-1]), weight.t()).view(x.shape[:-1] + (weight.shape[0],))
    if relu and bias is not None and x.dim() >= 2:
        x2 = x.reshape(-1, x.shape[-1])
        return torch._addmm_activation(bias, x2, weight.t(), use_gelu=False).reshape(x.shape[:-1] + (weight.shape[0],))
    y = F.linear(x, weight, bias)
    return F.relu_(y) if relu else y


def project_qk(x, w_q, b_q, w_k, b_k):
    """q and k projections of the same tensor as ONE library GEMM with stacked weights -> (q, k)."""
    if AG.needs_grad(x, w_q, b_q, w_k, b_k):
        return F.linear(x, w_q, b_q), F.linear(x, w_k, b_k)
    if _ops.linear_f16_ok(x, w_q):           # (the split weights are cached per parameter: no stacked copy)
        return _ops.linear_f16(x, w_q, b_q), _ops.linear_f16(x, w_k, b_k)
    qk = F.linear(x, torch.cat((w_q, w_k), 0), torch.cat((b_q, b_k), 0))
    C = w_q.shape[0]
    return qk[..., :C].contiguous(), qk[..., C:].contiguous()


def project_values_transposed(x, w_v, b_v):
    """Value projection emitted directly in the operand layout of the P.V MFMAs: (..., C, Mp) = W_v x^T + b with the keys
    zero-padded to the key stride (padded keys carry probability 0, their values never matter)."""
    M = x.shape[-2]
    Mp = _ops.key_stride(M)
    if LINEAR_STREAM and Mp == M and not AG.needs_grad(x, w_v, b_v) and x.is_contiguous() and _ops.linear_stream_ok(x, w_v):
        vt = _ops.linear_stream_transposed(x if x.dim() == 3 else x.unsqueeze(0), w_v, b_v)
        return vt if x.dim() == 3 else vt[0]
    xp = F.pad(x, (0, 0, 0, Mp - M)) if Mp != M else x
    C = w_v.shape[0]
    if xp.dim() == 2:
        return torch.addmm(b_v[:, None].expand(C, Mp), w_v, xp.t())
    A = xp.shape[0]
    return torch.baddbmm(b_v[None, :, None].expand(A, C, Mp), w_v[None].expand(A, C, C), xp.transpose(1, 2))


def add_layer_norm(hidden, residual, weight, bias, eps=1e-5, hidden_bias=None):
    """LayerNorm(hidden [+ hidden_bias] + residual) over the last dim (residual may broadcast over a leading anchor dim)."""
    if AG.needs_grad(hidden, residual, weight, bias, hidden_bias):
        fix = lambda h, r: r.contiguous() if r.shape == h.shape else r.reshape(r.shape[-2:])
        if hidden.shape[-1] % 4 == 0:        # hand-written backward (csrc/rowops.hip: add_ln_bwd_kernel); AG.add_layer_norm is its torch pin
            def bwd(g, needs, h, r, w, b, hb):
                dh, dr, dw, db, dhb = _ops.add_layer_norm_bwd(g, h, fix(h, r), w, eps, hb)
                return dh, dr.reshape(r.shape), dw, db, dhb
            return AG.hip_backward(lambda h, r, w, b, hb: _ops.add_layer_norm(h, fix(h, r), w, b, eps, hb), bwd, 'add_layer_norm',
                                   hidden, residual, weight, bias, hidden_bias)
        return AG.differentiable(lambda h, r, w, b, hb: _ops.add_layer_norm(h, fix(h, r), w, b, eps, hb),
                                 lambda h, r, w, b, hb: AG.add_layer_norm(h, r, w, b, hb, eps), 1, hidden, residual, weight, bias, hidden_bias)
    return _ops.add_layer_norm(hidden, residual, weight, bias, eps, hidden_bias)


def anchor_max(x, dim=1):
    """Maximum over the anchor axis (6 anchors: one HIP launch; anything else: torch)."""
    if AG.needs_grad(x):
        return x.amax(dim)
    return _ops.anchor_max(x, dim) if x.dim() == 3 else x.amax(dim)


def gather_rows_padded(x, idx):
    """x[idx] where idx == x.shape[0] addresses an implicit all-zero row."""
    if AG.needs_grad(x):
        def bwd(g, needs, x_, idx_):        # rows back onto their sources; the padding row n is dropped
            n = x_.shape[0]
            if _ops.TRAINING_DETERMINISTIC and g.is_cuda:      # order-independent fixed-point sums (index_add_: float atomics)
                return _ops.scatter_add_rows(g, idx_, n), None
            dx = g.new_zeros((n + 1,) + tuple(x_.shape[1:]))
            flat = torch.where((idx_ < 0) | (idx_ > n), torch.full_like(idx_, n), idx_).reshape(-1)
            dx.index_add_(0, flat, g.reshape((-1,) + tuple(x_.shape[1:])))
            return dx[:n], None
        return AG.hip_backward(_ops.gather_rows_padded, bwd, 'gather_rows_padded', x, idx)
    return _ops.gather_rows_padded(x, idx)


def neighbor_max_pool(x, idx):
    """max over the neighbour rows of x (P_s, ...) for each query (idx (P_q, NN), padding row = zeros)."""
    if AG.needs_grad(x):
        return AG.hip_backward(_ops.neighbor_max_pool, lambda g, needs, x_, idx_: (_ops.neighbor_max_pool_bwd(x_, idx_, g), None),
                               'neighbor_max_pool', x, idx)
    return _ops.neighbor_max_pool(x, idx)


def apply_transform(points, T):
    return points @ T[..., :3, :3].transpose(-1, -2) + T[..., None, :3, 3]


# ---------------------------------------------------------------------------------------------------------------------
# B: backbone ops
# ---------------------------------------------------------------------------------------------------------------------
class norm_segments:
    """Context: while active, GroupNorm inside the backbone normalises every pair of a stacked batch separately.  `stage_offsets[i]` =
    [0, end of pair 0, end of pair 1, ...] in stacked points of pyramid stage i; the backbone announces the stage each block runs at
    (`norm_segments.at_stage(i)`), so two stages that happen to hold the same number of points cannot be confused.  Thread-local (batches
    may be processed by several host threads)."""
    _tls = threading.local()

    def __init__(self, stage_offsets):
        self.stage_offsets = [list(o) for o in stage_offsets]

    def __enter__(self):
        self.prev = (getattr(norm_segments._tls, 'current', None), getattr(norm_segments._tls, 'stage', None))
        norm_segments._tls.current, norm_segments._tls.stage = self.stage_offsets, None
        return self

    def __exit__(self, *exc):
        norm_segments._tls.current, norm_segments._tls.stage = self.prev
        return False

    @staticmethod
    def at_stage(i, support=None):
        """Called by the backbone in front of the blocks of pyramid stage i (a no-op outside a norm_segments context).  A strided block
        also normalises its INPUT rows, which live at the finer stage `support`; a tensor belongs to the one of the two stages whose point
        count it has (grid subsampling only removes points per cloud: equal totals mean equal pair boundaries)."""
        norm_segments._tls.stage = None if i is None else ((i,) if support is None else (i, support))
        hook = getattr(norm_segments._tls, 'stage_hook', None)
        if hook is not None:
            hook(i)


def row_segments(x):
    """Row offsets [0, end of pair 0, ...] of the independently normalised row ranges of x (points x anchors flattened), or None for a
    single range: from the active norm_segments context and the stage the backbone announced."""
    ctx, stage = getattr(norm_segments._tls, 'current', None), getattr(norm_segments._tls, 'stage', None)
    if ctx is None or stage is None:
        return None
    offs = next((ctx[i] for i in stage if ctx[i][-1] == x.shape[0]), None)
    if offs is None:
        raise RuntimeError('group_norm_rows: %d rows inside a block of pyramid stage(s) %s holding %s stacked points'
                           % (x.shape[0], stage, [ctx[i][-1] for i in stage]))
    if len(offs) <= 2:
        return None
    mult = x.numel() // x.shape[-1] // x.shape[0]
    return [o * mult for o in offs]


# Pending forms (inference): a GroupNorm [+ LeakyReLU] reduced to its affine table and applied by whoever loads the tensor next
# (se3et_amd.ops.Pending; csrc/dense_norm.hip, csrc/rowops.hip: gn_chain_apply_kernel).
Pending = _ops.Pending
PENDING_NORM = True           # False: every GroupNorm as its own three launches (A/B runs, tools/micro)


def dense_norm_ok(x, weight, groups):
    raw = x.raw if isinstance(x, Pending) else x
    return PENDING_NORM and not AG.needs_grad(raw, weight) and _ops.dense_norm_ok(x, weight, groups)


@_hip
def dense_norm(x, weight, linear_bias, norm_weight, norm_bias, groups, eps):
    """Dense layer + the statistics of the GroupNorm that follows it -> Pending(raw = T(x) W^T, [affine table]) (slope 1: the caller sets
    the LeakyReLU slope of its block).  x: a tensor or a Pending."""
    segments = x.segments if isinstance(x, Pending) else row_segments(x)
    return _ops.dense_norm(x, weight, linear_bias, norm_weight, norm_bias, groups, eps, segments)


RECOMPUTE_TAIL = os.environ.get('SE3_RECOMPUTE_TAIL', '1') != '0'         # False: the round-3 block tail (unary2 / skip_conv store their raw output, one apply pass adds them: A/B runs)


@_hip
def dense_stats(x, weight, linear_bias, norm_weight, norm_bias, groups, eps):
    """The affine table of GroupNorm(T(x) W^T + bias) from a GEMM whose product is never stored (csrc/dense_norm.hip, statistics-only mode)."""
    segments = x.segments if isinstance(x, Pending) else row_segments(x)
    return _ops.dense_stats(x, weight, linear_bias, norm_weight, norm_bias, groups, eps, segments)


@_hip
def dense_residual(x, weight, affine, residual=None, shortcut=None, final_slope=1.0):
    """lrelu(GroupNorm(T(x) W^T + b) + R) with the GroupNorm as its table `affine` (dense_stats) and R a tensor, a shortcut layer
    (x2, weight2, affine2) evaluated in the same kernel, or nothing: the block's output is the only wide tensor written."""
    segments = x.segments if isinstance(x, Pending) else row_segments(x)
    return _ops.dense_residual(x, weight, affine, residual, shortcut, final_slope, segments)


@_hip
def norm_stats(x, norm_weight, norm_bias, groups, eps, slope, x_bias=None):
    """x (tensor, or Pending with one stage) -> x with one more pending GroupNorm + LeakyReLU(slope) stage."""
    pend = x if isinstance(x, Pending) else Pending(x, [], [], row_segments(x))
    return pend.then(_ops.group_norm_stats(pend, norm_weight, norm_bias, groups, eps, x_bias), slope)


@_hip
def norm_apply(x, residual=None, final_slope=1.0, blocked=False, union=False):
    """A Pending made concrete, optionally + residual (tensor or one-stage Pending) and a final LeakyReLU.  blocked=True: as the fused
    KPConv's gather layout (ops.BlockedFeatures; only kpconv_inter_so3 takes it)."""
    kind = 0
    if blocked and _ops.KPCONV_BLOCKED and x.raw.dim() == 3 and x.raw.shape[1] == 6:
        kind = 2 if (union and x.raw.shape[2] % 8 == 0) else (1 if x.raw.shape[2] % 16 == 0 else 0)
    return _ops.group_norm_apply(x, residual, final_slope, kind)


def group_norm_rows(x, weight, bias, groups, eps, leaky_slope=None, residual=None, x_bias=None, segments=None):
    """GroupNorm over (rows x channels-in-group) for x (..., C) with ALL leading dims pooled into the statistics
    (GroupNormEPN / kpconv GroupNorm), optionally `+ residual` then LeakyReLU, fused in one pass."""
    if segments is None:
        segments = row_segments(x)
    if AG.needs_grad(x, weight, bias, residual, x_bias):
        # hand-written backward (csrc/rowops.hip: gn_bwd_*); AG.group_norm_rows is its torch pin
        def bwd(g, needs, x_, w, b, r, xb):
            dx, dw, db, dr, dxb = _ops.group_norm_rows_bwd(g, x_, w, b, groups, eps, leaky_slope, r, xb, segments)
            return dx, dw, db, dr, dxb
        return AG.hip_backward(lambda x_, w, b, r, xb: _ops.group_norm_rows(x_, w, b, groups, eps, leaky_slope, r, xb, segments), bwd,
                               'group_norm_rows', x, weight, bias, residual, x_bias)
    return _ops.group_norm_rows(x, weight, bias, groups, eps, leaky_slope, residual, x_bias, segments)


def kpconv_takes_union(q_pts, s_pts, Cin, Cout):
    """True when kpconv_inter_so3 will run the union-staged kernel for this layer (a spatial order of q_pts is registered and the policy
    picks it): its input is then written in that kernel's row layout (ops.BlockedFeatures kind 2)."""
    return bool(_ops.KPCONV_UNION and q_pts.is_cuda and _ops._kpconv_union_pays(Cin, Cout, q_pts is s_pts) and _ops.point_order(q_pts) is not None)


def kpconv_inter_so3(x, q_pts, s_pts, idx, kernel_points, weights, kidx, ridx, sigma):
    if AG.needs_grad(x, weights):
        # hand-written backward (csrc/kpconv_so3.hip: kpconv_scatter_kernel + two library GEMMs); AG.kpconv_inter_so3 is its torch pin
        return AG.hip_backward(lambda x_, w: _ops.kpconv_inter_so3(x_, q_pts, s_pts, idx, kernel_points, w, kidx, ridx, sigma),
                               lambda g, needs, x_, w: _ops.kpconv_inter_so3_bwd(g, x_, q_pts, s_pts, idx, kernel_points, w, kidx, ridx, sigma,
                                                                                 need_x=needs[0], need_w=needs[1]),
                               'kpconv_inter_so3', x, weights)
    return _ops.kpconv_inter_so3(x, q_pts, s_pts, idx, kernel_points, weights, kidx, ridx, sigma)


# ---------------------------------------------------------------------------------------------------------------------
# G: geometric structure embedding
# ---------------------------------------------------------------------------------------------------------------------
def sinusoidal_embedding(idx, div_term):
    om = idx.reshape(-1, 1, 1) * div_term.view(1, -1, 1)
    return torch.cat((torch.sin(om), torch.cos(om)), 2).reshape(*idx.shape, 2 * div_term.numel())


def geometric_embedding(points, div_term, w_d, b_d, w_a, b_a, sigma_d, sigma_a, k, wigner_d1=None, dtype=torch.float32, knn=None, tables=None):
    """E (N, N, C) [and the equivariant embedding (A, N, N, 4) when the Wigner-D^1 table is given] in one kernel; dtype
    torch.bfloat16 stores E rounded to bf16 ('bf16 attention', BASELINE.json configs[2])."""
    if AG.needs_grad(w_d, b_d, w_a, b_a):
        if dtype != torch.float32:
            raise RuntimeError('geometric_embedding: training runs in float32')
        if knn is None:
            knn = _ops.knn3_stack(points.contiguous(), [points.shape[0]])
        hip = lambda wd, bd, wa, ba: _ops.geometric_embedding(points, div_term, wd, bd, wa, ba, sigma_d, sigma_a, k, wigner_d1, dtype, knn, tables)
        ref = lambda wd, bd, wa, ba: AG.geometric_embedding(points, div_term, wd, bd, wa, ba, knn, sigma_d, sigma_a)
        # hand-written backward: the kernel writes the GEMM operands of the four weight gradients (csrc/geo_embedding.hip); `ref` is its torch pin
        if not AG.needs_grad(points):
            # (tables validated at forward time stay valid in backward: the weights change in optimizer.step only)
            bwd = lambda g, needs, wd, bd, wa, ba: _ops.geometric_embedding_bwd(g, points, div_term, wd, bd, wa, ba, sigma_d, sigma_a, knn, tables)
            out = AG.hip_backward(hip, bwd, 'geometric_embedding', w_d, b_d, w_a, b_a)
            return out if wigner_d1 is None else (out[0], out[1].detach())
        if wigner_d1 is None:
            return AG.differentiable(hip, ref, 1, w_d, b_d, w_a, b_a)
        # the equivariant embedding has no learned inputs: second output of the kernel, constant for autograd
        emb, eq = AG.differentiable(hip, lambda *a: (ref(*a), points.new_zeros(1)), 2, w_d, b_d, w_a, b_a)
        return emb, eq.detach()
    return _ops.geometric_embedding(points, div_term, w_d, b_d, w_a, b_a, sigma_d, sigma_a, k, wigner_d1, dtype, knn, tables)


# ---------------------------------------------------------------------------------------------------------------------
# D: attention
# ---------------------------------------------------------------------------------------------------------------------
def rpe_attention(q, k, vt, emb, w_p, eq_emb, w_eq, num_heads, return_scores=False):
    """q ([A,] N, C), k ([A,] M, C) already projected, vt ([A,] C, Mp) = project_values_transposed; emb (N, M, C); eq_emb
    (A, N, M, 4) or None.  softmax_m((q.k + q.(W_p emb) + q.(W_eq eq_emb)) / sqrt(d)) v with the position terms folded onto
    the query side (q.(W e + b) = (W^T q).e + q.b; the q.b term is constant along m and cancels in the softmax).
    Convenience form (the layers use `rpe_self_attention_packed`, which gets q, k and the folded queries from ONE GEMM)."""
    if AG.needs_grad(q, k, vt, emb, w_p, w_eq):
        fwd = lambda q_, k_, v_, e_, wp, ee, we: rpe_attention(q_, k_, v_, e_, wp, ee, we, num_heads)[0]
        if ATTENTION_BACKWARD == 'explicit' and emb.dtype == torch.float32 and (q.dim() == 2 or q.shape[0] * num_heads <= 32):
            # hand-derived backward (se3et_amd/attention_bwd.py): logits recomputed by the forward's HIP kernel, batched library GEMMs
            hidden = AG.hip_backward(fwd, lambda g, needs, *t: ABW.rpe_attention_bwd(g, needs, *t, num_heads), 'rpe_attention',
                                     q, k, vt, emb, w_p, eq_emb, w_eq)
        else:
            hidden = AG.differentiable(fwd, lambda q_, k_, v_, e_, wp, ee, we: AG.rpe_attention(q_, k_, v_, e_, wp, ee, we, num_heads),
                                       1, q, k, vt, emb, w_p, eq_emb, w_eq)
        return hidden, None
    anchored = q.dim() == 3
    q3 = q if anchored else q.unsqueeze(0)
    A, N, C = q3.shape
    H = num_heads
    d = C // H
    qh = q3.reshape(A, N, H, d)
    qp = torch.einsum('anhd,hdc->anhc', qh, w_p.view(H, d, C)).reshape(A, N, H * C)
    qe = None
    if eq_emb is not None:      # qp and qe must be column blocks of one tensor (shared strides)
        both = torch.cat((qp, torch.einsum('anhd,hde->anhe', qh, w_eq.view(H, d, 4)).reshape(A, N, 4 * H)), -1)
        qp, qe = both[..., :H * C], both[..., H * C:]
    bias = _ops.rpe_bias(qp, qe, emb, eq_emb, H)
    out = _ops.attention(q3, k if anchored else k.unsqueeze(0), vt if anchored else vt.unsqueeze(0), bias, H, tag='rpe')
    scores = None
    if return_scores:        # diagnostic path: the product never needs the (A, H, N, M) tensor
        M = emb.shape[1]
        k3 = k if anchored else k.unsqueeze(0)
        sc = torch.einsum('anhd,amhd->ahnm', qh, k3.reshape(A, M, H, d)) + bias.view(A, H, N, -1)[..., :M]
        scores = torch.softmax(sc / math.sqrt(d), -1)
        scores = scores if anchored else scores[0]
    return (out if anchored else out[0]), scores


def compose_self_attention_weights(w_q, b_q, w_k, b_k, w_p, w_eq, num_heads):
    """Stacked projection [q | k | qp | qe] of RPE self attention as ONE linear layer.  qp = W_p^T q and qe = W_eq^T q are
    linear in x, so their weights compose with W_q: W_qp[(h, c), i] = sum_j W_q[h d + j, i] W_p[h d + j, c] (float64
    composition, cached by the caller per weight version).  Returns (weight (2C + HC [+ 4H], C), bias, column offsets)."""
    C = w_q.shape[0]
    H = num_heads
    d = C // H
    wq, bq, wp = w_q.detach().double(), b_q.detach().double(), w_p.detach().double()
    w_qp = torch.einsum('hji,hjc->hci', wq.view(H, d, C), wp.view(H, d, C)).reshape(H * C, C)
    b_qp = torch.einsum('hj,hjc->hc', bq.view(H, d), wp.view(H, d, C)).reshape(H * C)
    ws, bs = [wq, w_k.detach().double(), w_qp], [bq, b_k.detach().double(), b_qp]
    offs = {'q': 0, 'k': C, 'qp': 2 * C, 'qe': None}
    if w_eq is not None:
        we = w_eq.detach().double()
        ws.append(torch.einsum('hji,hje->hei', wq.view(H, d, C), we.view(H, d, 4)).reshape(4 * H, C))
        bs.append(torch.einsum('hj,hje->he', bq.view(H, d), we.view(H, d, 4)).reshape(4 * H))
        offs['qe'] = 2 * C + H * C
    return torch.cat(ws, 0).float().contiguous(), torch.cat(bs, 0).float().contiguous(), offs


def pack_rows(xs, multiple=32):
    """([A,] N_i, C) tensors -> one zero-padded ([A,] sum ceil_mult(N_i), C) tensor + the row offset of every segment."""
    starts, total = [], 0
    for x in xs:
        starts.append(total)
        total += (x.shape[-2] + multiple - 1) // multiple * multiple
    packed = torch.zeros(xs[0].shape[:-2] + (total, xs[0].shape[-1]), dtype=xs[0].dtype, device=xs[0].device)
    for x, s0 in zip(xs, starts):
        packed[..., s0:s0 + x.shape[-2], :] = x
    return packed, starts


def rpe_self_attention_packed(x, starts, lengths, embs, eq_embs, w_stack, b_stack, offs, w_v, b_v, num_heads):
    """RPE self attention of several clouds at once.  x ([A,] R, C): the clouds' rows packed at `starts` (multiples of 32) with
    `lengths`; ONE stacked GEMM yields q, k and the folded queries of all clouds, ONE GEMM the transposed values; the two
    attention kernels run once for all clouds (stack mode).  Returns hidden ([A,] R, C) (padding rows are zero)."""
    H = num_heads
    C = x.shape[-1]
    x3 = x if x.dim() == 3 else x.unsqueeze(0)
    proj = linear(x3, w_stack, b_stack)                                    # (A, R, 2C + HC [+ 4H])
    vt = project_values_transposed(x3, w_v, b_v)                           # (A, C, R): R is a multiple of 32 (packed rows)
    hidden = torch.zeros_like(x3)
    _ops.rpe_self_attention_stack(proj, offs, vt, embs, eq_embs, starts, lengths, H, hidden)
    return hidden if x.dim() == 3 else hidden[0]


def cross_attention(q, k, vt, num_heads):
    """q (N, C), k (M, C), vt (C, Mp) or (A, C, Mp) -> (N, C) or (A, N, C)."""
    if AG.needs_grad(q, k, vt):
        if ATTENTION_BACKWARD == 'explicit':
            return AG.hip_backward(lambda q_, k_, v_: _ops.cross_attention(q_, k_, v_, num_heads),
                                   lambda g, needs, *t: ABW.cross_attention_bwd(g, needs, *t, num_heads), 'cross_attention', q, k, vt)
        return AG.differentiable(lambda q_, k_, v_: _ops.cross_attention(q_, k_, v_, num_heads),
                                 lambda q_, k_, v_: AG.cross_attention(q_, k_, v_, num_heads), 1, q, k, vt)
    return _ops.cross_attention(q, k, vt, num_heads)


def cross_attention_eq(q, k, vt, num_heads, mode, trace_idx):
    """q (A, N, C), k (A, M, C), vt (A, C, Mp).  Returns (hidden (A, N, C), weights, mix): weights = g/sum_e g (A, A) for
    'a_soft', w (R,) for 'r_soft'; mix (A, A) = the anchor-pair weights actually applied (for r_soft the 24 rotation weights
    collapsed onto anchor pairs, mix[a, e] = sum_{r: trace[r, a] = e} w[r])."""
    if AG.needs_grad(q, k, vt):
        if ATTENTION_BACKWARD == 'explicit':
            return ABW.cross_attention_eq(q, k, vt, num_heads, mode, trace_idx)
        return AG.differentiable(lambda q_, k_, v_, t_: _ops.cross_attention_eq(q_, k_, v_, num_heads, mode, t_),
                                 lambda q_, k_, v_, t_: AG.cross_attention_eq(q_, k_, v_, t_, num_heads, mode), 3, q, k, vt, trace_idx)
    return _ops.cross_attention_eq(q, k, vt, num_heads, mode, trace_idx)


def rotation_weighted_permute(feats, mix):
    """sum_r w[r] feats[:, trace[r, a]] for feats (B, A, N, C) with the rotation sum collapsed onto mix (A, A)."""
    return torch.einsum('ae,benc->banc', mix, feats)


# ---------------------------------------------------------------------------------------------------------------------
# E: matching
# ---------------------------------------------------------------------------------------------------------------------
def superpoint_scores(ref_feats, src_feats, dual_normalization=True):
    return _ops.superpoint_scores(ref_feats, src_feats, dual_normalization)


def log_optimal_transport(scores, row_masks, col_masks, alpha, num_iterations, inf):
    if AG.needs_grad(scores, alpha):
        # hand-written backward (csrc/sinkhorn.hip: sinkhorn_bwd_kernel) where the dual history fits in LDS; AG.log_optimal_transport is its torch pin
        if num_iterations * (scores.shape[1] + scores.shape[2] + 2) * 4 <= 150 * 1024 and max(scores.shape[1:]) <= 143:
            return AG.hip_backward(lambda s_, a_, rm, cm: _ops.log_optimal_transport(s_, rm, cm, a_, num_iterations, inf),
                                   lambda g, needs, s_, a_, rm, cm: _ops.log_optimal_transport_bwd(g, s_, rm, cm, a_, num_iterations, inf) + (None, None),
                                   'log_optimal_transport', scores, alpha, row_masks, col_masks)
        return AG.differentiable(lambda s_, a_, rm, cm: _ops.log_optimal_transport(s_, rm, cm, a_, num_iterations, inf),
                                 lambda s_, a_, rm, cm: AG.log_optimal_transport(s_, a_, rm, cm, num_iterations, inf),
                                 1, scores, alpha, row_masks, col_masks)
    return _ops.log_optimal_transport(scores, row_masks, col_masks, alpha, num_iterations, inf)


# ---------------------------------------------------------------------------------------------------------------------
# F1: registration (device-side Kabsch, csrc/registration.hip)
# ---------------------------------------------------------------------------------------------------------------------
def mutual_topk_mask(scores, row_masks, col_masks, k, threshold):
    return _ops.mutual_topk_mask(scores, row_masks, col_masks, k, threshold)


def weighted_procrustes(src, ref, scores, offsets, gate_transform=None, gate_radius=0.0, eps=1e-5):
    return _ops.weighted_procrustes(src, ref, scores, offsets, gate_transform, gate_radius, eps)


def count_inliers(src, ref, transforms, radius, range_begin=None, range_end=None):
    return _ops.count_inliers(src, ref, transforms, radius, range_begin, range_end)
