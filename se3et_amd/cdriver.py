"""Host side of the C-issued transformer (csrc/transformer_driver.hip: se3_transformer_forward).

`transformer_forward(gt, X, PA, embs, eqs)` builds the plan -- the layers' weights as f16 pieces (cached per weight version like every other
dense layer), the packed-row layout, the embeddings of the clouds -- and makes ONE library call that issues the ~130 launches of the ten
blocks and of out_proj.  Results equal se3et_amd.batched.transformer_pairs (the same kernels with the same operands in the same order);
`supported(gt)` says whether a model's block list / sizes are covered (anything else keeps the Python schedule)."""
import ctypes
import os

import torch

from . import ops as _ops
from ._lib import check, lib

MAX_BLOCKS, MAX_BATCH = 16, 32
_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class Linear(ctypes.Structure):
    _fields_ = [('pieces', _vp), ('bias', _vp), ('in_features', _i32), ('out_features', _i32)]


class Layer(ctypes.Structure):
    _fields_ = [('type', _i32), ('off_q', _i32), ('off_k', _i32), ('off_qp', _i32), ('off_qe', _i32), ('stack', Linear), ('q', Linear),
                ('k', Linear), ('v', Linear), ('out', Linear), ('out_pieces_g2', _vp), ('out_pieces_g3', _vp), ('ln1_w', _vp), ('ln1_b', _vp),
                ('ln1_eps', _f32), ('expand', Linear), ('squeeze', Linear), ('ln2_w', _vp), ('ln2_b', _vp), ('ln2_eps', _f32), ('trace_idx', _vp),
                ('num_rotations', _i32)]


class Plan(ctypes.Structure):
    _fields_ = [('A', _i32), ('C', _i32), ('H', _i32), ('num_blocks', _i32), ('num_pairs', _i32), ('emb_bf16', _i32), ('layers', Layer * MAX_BLOCKS),
                ('rc_expand', Linear), ('rc_squeeze', Linear), ('rc_ln_w', _vp), ('rc_ln_b', _vp), ('rc_ln_eps', _f32), ('out_proj', Linear),
                ('starts', _i64 * MAX_BATCH), ('lengths', _i64 * MAX_BATCH), ('rows0', _i64), ('rows', _i64), ('emb', _vp * MAX_BATCH),
                ('eq', _vp * MAX_BATCH)]


def check_layout():
    """The ctypes mirror against the loaded library (se3_transformer_plan_layout): sizes of the three structs and three field offsets."""
    got = (ctypes.c_size_t * 6)()
    lib().se3_transformer_plan_layout(got)
    mine = (ctypes.sizeof(Linear), ctypes.sizeof(Layer), ctypes.sizeof(Plan), Plan.layers.offset, Plan.starts.offset, Plan.emb.offset)
    if tuple(got) != mine:
        raise RuntimeError('se3et_amd.cdriver: plan structs %s do not match the library\'s %s' % (mine, tuple(got)))


_layout_checked = False
_TYPES = {'self': 0, 'self_eq': 1, 'cross': 2, 'cross_a_soft': 3, 'cross_r_soft': 4}
ENABLED = os.environ.get('SE3_CDRIVER', '1') != '0'            # False: the Python schedule of se3et_amd.batched (A/B runs, tests)


def _schedule_ok(blocks, has_rotcompress):
    """The scheduler state of se3_transformer_forward (csrc/transformer_driver.hip: X_is_eq / Xeq) run over a block list: False wherever the
    driver would return SE3_ERR_UNSUPPORTED / SE3_ERR_INVALID_ARG -- a plain cross block on anchor features, anchor values without a
    self_eq block in front, an equivariant cross block on invariant features, eq2inv without the rotcompress layer, a list that ends on
    anchor features."""
    x_is_eq, has_xeq = True, False
    types = [_TYPES[b] for b in blocks]
    for i, t in enumerate(types):
        nxt = types[i + 1] if i + 1 < len(types) else -1
        prev = types[i - 1] if i > 0 else -1
        if t <= 1:
            anchors = has_xeq or x_is_eq
            if t == 1 and nxt == 2:
                has_xeq, x_is_eq = True, False
            else:
                x_is_eq = anchors
        elif t == 2:
            if x_is_eq:
                return False
            if nxt == 1 or (nxt < 0 and prev == 1):
                if not has_xeq:
                    return False
        else:
            if not x_is_eq:
                return False
            if t == 4 and 0 <= nxt < 3 and nxt != 1:
                if not has_rotcompress:
                    return False
                x_is_eq, has_xeq = False, False
    return not x_is_eq


def supported(gt):
    """The block lists of the SE3ET experiments on the kernels' sizes: every block type known, 6 anchors, channels a multiple of 32 and of
    the head count, and a block order the driver's scheduler accepts (_schedule_ok mirrors its SE3_REQUIREs: anything it would refuse keeps
    the Python schedule instead of surfacing as a RuntimeError)."""
    tr = gt.transformer
    C = gt.in_proj.out_features
    if not ENABLED or len(tr.blocks) > MAX_BLOCKS or any(b not in _TYPES for b in tr.blocks) or gt.na != 6 or C % 32:
        return False
    if C % tr.layers[0].attention.attention.num_heads:
        return False
    if tr.blocks[0] not in ('self_eq', 'cross_a_soft', 'cross_r_soft'):      # (the input is the (A, rows, C) in_proj of anchor features)
        return False
    if any(b in ('cross_a_soft', 'cross_r_soft') for b in tr.blocks) and C not in (128, 256):
        return False                      # (the Gram-matrix statistics kernel of the equivariant cross attention: 128 or 256 channels)
    return _schedule_ok(tr.blocks, hasattr(tr, 'rotcompress'))


def _linear(weight, bias, stream, keep):
    """se3_linear_t of a dense layer: the cached f16 pieces of its weight + the bias pointer."""
    Wp = _ops._linear_weight_pieces(weight, stream)
    keep.append(Wp)
    b = None
    if bias is not None:
        b = bias.detach()
        keep.append(b)
    return Linear(Wp.data_ptr(), b.data_ptr() if b is not None else None, int(weight.shape[1]), int(weight.shape[0]))


def _static_plan(gt, stream):
    """The weight part of the plan, rebuilt when any parameter's version changes (an optimizer step, load_state_dict) or when
    ops.clear_weight_caches() / ops.validate_weight_caches() dropped cached pieces (ops.CACHE_EPOCH: writes behind the version counter):
    pointers into the per-weight piece caches of se3et_amd.ops (which own the device memory)."""
    tr = gt.transformer
    params = list(gt.parameters())
    key = (_ops.CACHE_EPOCH[0], tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
    plans = gt.__dict__.setdefault('_cdriver_plans', {})            # one per launch stream (the pieces' cross-stream waits happen at build time)
    hit = plans.get(stream.value)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    keep = []
    plan = Plan()
    C = gt.in_proj.out_features
    plan.A, plan.C, plan.H = gt.na, C, tr.layers[0].attention.attention.num_heads
    plan.num_blocks = len(tr.blocks)
    for i, (block, layer) in enumerate(zip(tr.blocks, tr.layers)):
        L = plan.layers[i]
        L.type = _TYPES[block]
        al, att, out = layer.attention, layer.attention.attention, layer.output
        if L.type <= 1:
            w, b, offs = att.stacked_projection()
            keep += [w, b]
            L.stack = _linear(w, b, stream, keep)
            L.off_q, L.off_k, L.off_qp = offs['q'], offs['k'], offs['qp']
            L.off_qe = offs['qe'] if offs['qe'] is not None else -1
        else:
            L.q = _linear(att.proj_q.weight, att.proj_q.bias, stream, keep)
            L.k = _linear(att.proj_k.weight, att.proj_k.bias, stream, keep)
            L.off_qe = -1
        L.v = _linear(att.proj_v.weight, att.proj_v.bias, stream, keep)
        L.out = _linear(al.linear.weight, al.linear.bias, stream, keep)
        if L.type >= 3:
            for g, name in ((2, 'out_pieces_g2'), (3, 'out_pieces_g3')):
                Wg = _ops.stacked_weight(al.linear.weight, g)
                Wp = _ops._linear_weight_pieces(Wg, stream)
                keep += [Wg, Wp]
                setattr(L, name, Wp.data_ptr())
            t = att.trace_idx_ori.detach().contiguous()
            keep.append(t)
            L.trace_idx, L.num_rotations = t.data_ptr(), int(t.shape[0])
        for nm, ln in (('ln1', al.norm), ('ln2', out.norm)):
            setattr(L, nm + '_w', ln.weight.data_ptr())
            setattr(L, nm + '_b', ln.bias.data_ptr())
            setattr(L, nm + '_eps', float(ln.eps))
        L.expand = _linear(out.expand.weight, out.expand.bias, stream, keep)
        L.squeeze = _linear(out.squeeze.weight, out.squeeze.bias, stream, keep)
    if hasattr(tr, 'rotcompress'):
        rc = tr.rotcompress
        plan.rc_expand = _linear(rc.expand.weight, rc.expand.bias, stream, keep)
        plan.rc_squeeze = _linear(rc.squeeze.weight, rc.squeeze.bias, stream, keep)
        plan.rc_ln_w, plan.rc_ln_b, plan.rc_ln_eps = rc.norm.weight.data_ptr(), rc.norm.bias.data_ptr(), float(rc.norm.eps)
    plan.out_proj = _linear(gt.out_proj.weight, gt.out_proj.bias, stream, keep)
    if len(plans) > 16:
        plans.clear()
    plans[stream.value] = (key, plan, keep)
    return plan, keep


_ws = {}


def transformer_forward(gt, X, PA, R0, embs, eqs):
    """X (A, R, C) packed in_proj features (refs of all pairs, then srcs), PA the packing (se3et_amd.batched._Packed over all 2 B clouds in
    that order), R0 the packed rows of the refs, embs / eqs the clouds' embeddings in the same order (eqs entries None for models without
    the equivariant embedding).  -> (R, C_out) packed output rows of out_proj."""
    global _layout_checked
    if not _layout_checked:
        check_layout()
        _layout_checked = True
    stream = _ops._stream()
    static, keep = _static_plan(gt, stream)
    plan = Plan.from_buffer_copy(static)
    n = len(PA.lengths)
    if n % 2 or n > MAX_BATCH:
        raise RuntimeError('transformer_forward: %d clouds' % n)
    plan.num_pairs = n // 2
    plan.rows0, plan.rows = int(R0), int(PA.rows)
    plan.emb_bf16 = 1 if embs[0].dtype == torch.bfloat16 else 0
    for c in range(n):
        plan.starts[c], plan.lengths[c] = int(PA.starts[c]), int(PA.lengths[c])
        plan.emb[c] = embs[c].data_ptr()
        plan.eq[c] = eqs[c].data_ptr() if eqs[c] is not None else None
    X = _ops._req(X, torch.float32, 'X', 3)
    if tuple(X.shape) != (plan.A, plan.rows, plan.C):
        raise RuntimeError('transformer_forward: X %s for (%d, %d, %d)' % (tuple(X.shape), plan.A, plan.rows, plan.C))
    nbytes = lib().se3_transformer_workspace_bytes(ctypes.byref(plan))
    key = (X.device, stream.value)
    ws = _ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((int(nbytes * 1.1) + 256,), dtype=torch.uint8, device=X.device)
        _ws[key] = ws
    base = (ws.data_ptr() + 255) // 256 * 256
    out = torch.empty((plan.rows, gt.out_proj.out_features), dtype=torch.float32, device=X.device)
    check(lib().se3_transformer_forward(ctypes.byref(plan), X.data_ptr(), out.data_ptr(), base, ws.numel() - (base - ws.data_ptr()), stream),
          'se3_transformer_forward')
    return out
