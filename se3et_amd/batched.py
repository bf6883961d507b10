"""Several registration pairs per forward.

The reference runs one pair per `model.forward` (batch_size = 1, experiments/se3ete.3dmatch/model.py:79-227).  Here the clouds of
B pairs are stacked -- ref0, src0, ref1, src1, ... -- through the pyramid and the backbone (every op there is index based; GroupNorm
keeps one set of statistics per pair, se3_group_norm_segments_fwd), and packed row-wise through the transformer (all dense layers,
LayerNorms and both attention kernels run once for all 2 B clouds).  Results per pair are those of the single-pair forward: the
per-pair neighbour-table widths of the reference are reproduced (se3et_amd.data), GroupNorm / cross attention / matching never mix
pairs.  Launches per pair drop roughly with 1 / B, which is what bounds the single-pair path (≈730 launches, host ≈ GPU ≈ 10 ms).

Uses the parameters of an ordinary `se3et_amd.model.SE3ET`; nothing here is learned."""
import torch
import torch.nn.functional as F

from . import functional as SF
from . import cdriver as _cdriver
from . import ops as _ops
from .modules.transformer import _block_is_eq


def _offsets(lengths):
    out = [0]
    for n in lengths:
        out.append(out[-1] + int(n))
    return out


import contextlib as _contextlib
import os as _os
import threading as _threading

# Sections of the forward that the batches in flight (one host thread + HIP stream each) run ONE AFTER THE OTHER on the GPU: a section waits
# for the event behind the launches of the same section of the batch before it -- kernels of the same kind (the HBM-bound logits kernels,
# the register-full KPConv / dense kernels) then do not time-slice against each other, while different sections of different batches still
# overlap.  SE3_CHAIN = comma list of section names ('transformer', 'backbone'), '' = none.
CHAINED_SECTIONS = set(filter(None, _os.environ.get('SE3_CHAIN', 'transformer,backbone').split(',')))      # (round 6: both, each on its own chain -- profiles/r06_frac_vs_overlap.txt)
_chains = {}
_chains_guard = _threading.Lock()
SCHEDULE = 'environment'


def set_schedule(name):
    """How the sections of the batches in flight may overlap on the GPU (call between forwards, with every stream synchronised):
      'throughput'  backbone sections follow one another, transformer sections follow one another, a backbone beside a transformer is fine
                    (the library's default: 487-493 pairs/s at 16 pairs x 3 in flight, the RPE self-attention at 0.35-0.37 of HBM in that crowd)
      'roofline'    additionally the two COARSEST backbone stages (their wide dense / KPConv kernels) never run beside another batch's
                    transformer: the RPE kernels keep >= 0.40 of HBM while the other batches run (0.45-0.47; with the coarsest stage alone
                    0.41-0.44), at 455-458 pairs/s (round 6, profiles/r06_frac_vs_overlap.txt; bench.py's default)
      'exclusive'   no backbone stage beside a transformer: 0.49-0.51 (the kernels' own rate), 434-445 pairs/s"""
    global SCHEDULE, BACKBONE_SPLIT_STAGE, SHARED_CHAIN
    table = {'throughput': (None, {'transformer', 'backbone'}, {}),
             'roofline': (2, {'transformer', 'backbone', 'backbone_fine'}, {'transformer': 'transformer+backbone', 'backbone': 'transformer+backbone'}),
             'exclusive': (None, {'transformer', 'backbone'}, {'transformer': 'transformer+backbone', 'backbone': 'transformer+backbone'})}
    if name not in table:
        raise ValueError('schedule %r (one of %s)' % (name, sorted(table)))
    split, sections, groups = table[name]
    with _chains_guard:
        BACKBONE_SPLIT_STAGE, SHARED_CHAIN = split, False
        CHAINED_SECTIONS.clear()
        CHAINED_SECTIONS.update(sections)
        _CHAIN_OF.clear()
        _CHAIN_OF.update(groups)
        _chains.clear()
        SCHEDULE = name
# SE3_CHAIN_PRIORITY = comma list of chained sections that run on ONE shared high-priority stream (round 5): the section's kernels -- the
# transformer's HBM-bound RPE calls -- are dispatched ahead of the other batches' backbone / matching kernels instead of time-slicing with
# them; the caller's stream waits for the section and goes on.  Tensors the section returns are allocated on that stream: the caller marks
# them as used on its own (`adopt`).
PRIORITY_SECTIONS = set(filter(None, _os.environ.get('SE3_CHAIN_PRIORITY', '').split(',')))
# SE3_CHAIN_SHARED=1: ONE chain for all chained sections (a batch's backbone and another batch's transformer then never overlap either);
# SE3_CHAIN_GROUPS='transformer+backbone,tail': sections joined by '+' share a chain (round 6: the in-region roofline of the RPE kernels
# against the overlap that carries the throughput)
SHARED_CHAIN = _os.environ.get('SE3_CHAIN_SHARED', '0') == '1'
_LOCK_WAIT = {} if _os.environ.get('SE3_CHAIN_LOCK_STATS') == '1' else None
BACKBONE_SPLIT_STAGE = int(_os.environ['SE3_BACKBONE_SPLIT']) if _os.environ.get('SE3_BACKBONE_SPLIT') else None
_CHAIN_OF = {}
for _grp in filter(None, _os.environ.get('SE3_CHAIN_GROUPS', '').split(',')):
    for _sec in _grp.split('+'):
        _CHAIN_OF[_sec] = _grp
        CHAINED_SECTIONS.add(_sec)
_priority_streams = {}


class _Section:
    def __init__(self, caller, stream):
        self.caller, self.stream = caller, stream

    def adopt(self, *tensors):
        """Outputs of the section that the caller's stream goes on to use."""
        if self.stream is not None:
            for t in tensors:
                if torch.is_tensor(t):
                    t.record_stream(self.caller)
        return tensors[0] if len(tensors) == 1 else tensors


@_contextlib.contextmanager
def _chained(section):
    if section not in CHAINED_SECTIONS or not torch.cuda.is_available():
        yield _Section(None, None)
        return
    with _chains_guard:
        st = _chains.setdefault('*' if SHARED_CHAIN else _CHAIN_OF.get(section, section), [_threading.Lock(), None])
    if _LOCK_WAIT is not None:                    # (SE3_CHAIN_LOCK_STATS=1: host time spent waiting for a chain's lock, per chain)
        import time as _time
        t0 = _time.perf_counter()
        st[0].acquire()
        with _chains_guard:
            key = _CHAIN_OF.get(section, section)
            _LOCK_WAIT[key] = _LOCK_WAIT.get(key, 0.0) + _time.perf_counter() - t0
        st[0].release()
    with st[0]:
        cur = torch.cuda.current_stream()
        if section in PRIORITY_SECTIONS:
            key = (section, cur.device.index)
            hp = _priority_streams.get(key)
            if hp is None:
                hp = _priority_streams[key] = torch.cuda.Stream(device=cur.device, priority=-1)
            ev_in = torch.cuda.Event()
            ev_in.record(cur)
            hp.wait_event(ev_in)                    # (sections follow each other on hp by themselves)
            try:
                with torch.cuda.stream(hp):
                    yield _Section(cur, hp)
            finally:
                ev = torch.cuda.Event()
                ev.record(hp)
                cur.wait_event(ev)
            return
        if st[1] is not None:
            cur.wait_event(st[1])
        try:
            yield _Section(None, None)
        finally:
            ev = torch.cuda.Event()
            ev.record(cur)
            st[1] = ev


_pack_index_cache = {}     # (cloud order, cloud offsets, device) -> destination row of every stacked row (device tensor shared by the host threads)


class _Packed:
    """Rows of several clouds in one ([A,] R, C) tensor; cloud c = rows starts[c] .. + lengths[c] (starts multiples of 32)."""

    def __init__(self, lengths):
        self.lengths = [int(n) for n in lengths]
        self.starts, total = [], 0
        for n in self.lengths:
            self.starts.append(total)
            total += (n + 31) // 32 * 32
        self.rows = total

    def pack(self, xs):
        out = torch.zeros(xs[0].shape[:-2] + (self.rows, xs[0].shape[-1]), dtype=xs[0].dtype, device=xs[0].device)
        for x, s0 in zip(xs, self.starts):
            out[..., s0:s0 + x.shape[-2], :] = x
        return out

    def pack_rows(self, x, order, offs):
        """pack([x[..., offs[c]:offs[c + 1], :] for c in order]) as ONE scatter of rows (a copy per cloud otherwise: 16 launches for 8 pairs):
        x ([A,] P, C) holds the clouds back to back, cloud order[i] goes to packed rows starts[i] .. + lengths[i]."""
        key = (tuple(order), tuple(offs), str(x.device))
        dst = _pack_index_cache.get(key)
        if dst is None:
            rows = [0] * offs[-1]
            for i, c in enumerate(order):
                for j in range(offs[c + 1] - offs[c]):
                    rows[offs[c] + j] = self.starts[i] + j
            dst = SF.shared_tensors(_ops.to_device(rows, torch.int64, x.device))
            if len(_pack_index_cache) > 64:
                _pack_index_cache.clear()
            _pack_index_cache[key] = dst
        out = torch.zeros(x.shape[:-2] + (self.rows, x.shape[-1]), dtype=x.dtype, device=x.device)
        return out.index_copy_(x.dim() - 2, dst.get()[0], x)

    def unpack(self, x):
        return [x[..., s0:s0 + n, :] for s0, n in zip(self.starts, self.lengths)]


def _cross_plain(layer, Pq, Pk, xq, xk, xv):
    """One direction of a 'cross' block for all pairs: queries xq ([Rq, C]), keys from xk (Rk, C), values from xv ((Rk, C) or
    (A, Rk, C): per-anchor values share the invariant scores).  Returns the layer output in the packing of the queries."""
    att = layer.attention.attention
    H = att.num_heads
    q = SF.linear(xq, att.proj_q.weight, att.proj_q.bias)
    k = SF.linear(xk, att.proj_k.weight, att.proj_k.bias)
    C = q.shape[-1]
    w_v, b_v = att.proj_v.weight, att.proj_v.bias
    vt = SF.project_values_transposed(xv.contiguous(), w_v, b_v)           # (C, Rk) or (A, C, Rk): Rk a multiple of 32 (packed rows)
    if xv.dim() == 2:
        hidden = torch.zeros((Pq.rows, C), dtype=torch.float32, device=q.device)
    else:
        A = xv.shape[0]
        hidden = torch.zeros((A, Pq.rows, C), dtype=torch.float32, device=q.device)
    _ops.attention_stack(q, k, vt, None, None, Pq.starts, Pq.lengths, Pk.starts, Pk.lengths, H, hidden)
    al = layer.attention
    hidden = SF.linear(hidden, al.linear.weight)
    res = xq if hidden.dim() == xq.dim() else xq.unsqueeze(0)
    hidden = SF.add_layer_norm(hidden, res, al.norm.weight, al.norm.bias, al.norm.eps, hidden_bias=al.linear.bias)
    return layer.output(hidden)


def _cross_eq(layer, Pq, Pk, xq, xk):
    """One direction of a 'cross_a_soft' / 'cross_r_soft' block: dense layers on the packed rows of all pairs, the anchor-pair
    statistics and the weighted attention per pair.  Returns (output (A, Rq, C), per-pair mixing matrices)."""
    att = layer.attention.attention
    H = att.num_heads
    q = SF.linear(xq, att.proj_q.weight, att.proj_q.bias)
    k = SF.linear(xk, att.proj_k.weight, att.proj_k.bias)
    A, C = xk.shape[0], q.shape[-1]
    w_v, b_v = att.proj_v.weight, att.proj_v.bias
    vt = SF.project_values_transposed(xk.contiguous(), w_v, b_v)
    # few pairs: the key anchors divided over G workgroups whose partial results lie side by side along the channels; the output
    # projection below adds them (its weight repeated G times along the input dimension)
    G = _ops.cross_eq_groups(A, Pq.lengths, H, C, Pk.starts, vt) if q.is_contiguous() and k.is_contiguous() else 1
    hidden = torch.zeros((A, q.shape[1], G * C), dtype=torch.float32, device=q.device)
    mixes, _ = _ops.cross_attention_eq_stack(q, k, vt, Pq.starts, Pq.lengths, Pk.starts, Pk.lengths, H, att.attn_mode,
                                             att.trace_idx_ori, hidden, groups=G)
    al = layer.attention
    hidden = SF.linear(hidden, al.linear.weight if G == 1 else _ops.stacked_weight(al.linear.weight, G))
    hidden = SF.add_layer_norm(hidden, xq, al.norm.weight, al.norm.bias, al.norm.eps, hidden_bias=al.linear.bias)
    return layer.output(hidden), mixes


def transformer_pairs(gt, points_c, lengths_c, feats_c, packed=False):
    """Batched GeometricTransformer.forward: points_c (P, 3) / feats_c (P, A, C_in) stacked superpoints of 2 B clouds with
    `lengths_c`.  Returns (ref_feats [B tensors (N_i, C_out)], src_feats [B tensors]), or with packed=True the packed rows
    (R, C_out) and their _Packed layout (clouds ref0 .. ref(B-1), src0 .. src(B-1))."""
    offs = _offsets(lengths_c)
    clouds = [points_c[offs[c]:offs[c + 1]] for c in range(len(lengths_c))]
    B = len(clouds) // 2
    emb_mod = gt.embedding
    embs, eqs = [], []
    knn_all = _ops.knn3_stack(points_c, lengths_c)           # 3 nearest other superpoints, all clouds in one launch
    tabs = _ops.embedding_tables(emb_mod.embedding.div_term, emb_mod.proj_d.weight, emb_mod.proj_d.bias, emb_mod.proj_a.weight,
                                 emb_mod.proj_a.bias, emb_mod.sigma_a)      # validated against the weights once per forward
    for c, pts in enumerate(clouds):
        knn = knn_all[offs[c]:offs[c + 1]]
        args = (pts, emb_mod.embedding.div_term, emb_mod.proj_d.weight, emb_mod.proj_d.bias, emb_mod.proj_a.weight,
                emb_mod.proj_a.bias, emb_mod.sigma_d, emb_mod.sigma_a, emb_mod.angle_k)
        if gt.n_level_equiv > 0:
            e, q = SF.geometric_embedding(*args, wigner_d1=emb_mod.anchors_wignerD[1], dtype=emb_mod.embedding_dtype, knn=knn, tables=tabs)
        else:
            e, q = SF.geometric_embedding(*args, dtype=emb_mod.embedding_dtype, knn=knn, tables=tabs), None
        embs.append(e)
        eqs.append(q)
    # packing: all refs first, then all srcs, so that both halves are contiguous row ranges of one tensor
    order = [2 * i for i in range(B)] + [2 * i + 1 for i in range(B)]
    P0 = _Packed([lengths_c[2 * i] for i in range(B)])
    P1 = _Packed([lengths_c[2 * i + 1] for i in range(B)])
    PA = _Packed([lengths_c[c] for c in order])             # P0 followed by P1 (both row counts are multiples of 32)
    R0 = P0.rows
    x = SF.linear(feats_c, gt.in_proj.weight, gt.in_proj.bias).transpose(0, 1)                      # (A, P, C): a view, packed (copied) below
    X = PA.pack_rows(x, order, offs)                                                                 # (A, R, C)
    embs_o, eqs_o = [embs[c] for c in order], [eqs[c] for c in order]

    if _cdriver.supported(gt) and not torch.is_grad_enabled():
        # every launch of the ten blocks and of out_proj from ONE library call (csrc/transformer_driver.hip): the same kernels with the same
        # operands as the schedule below, no interpreter between them
        with _chained('transformer') as sec:
            X = sec.adopt(_cdriver.transformer_forward(gt, X.contiguous(), PA, R0, embs_o, eqs_o))
        if packed:
            return X, PA
        outs = PA.unpack(X)
        return outs[:B], outs[B:]
    tr = gt.transformer
    blocks, layers = tr.blocks, tr.layers
    X_eq = None                      # anchor features (A, R, C) kept next to their anchor-max X (R, C) inside 'cross' runs
    mixes0 = None
    for i, block in enumerate(blocks):
        layer = layers[i]
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        if 'self' in block:
            eq = block == 'self_eq'
            src = X_eq if X_eq is not None else X
            y = layer.attention.forward_packed(src, PA.starts, PA.lengths, embs_o, eqs_o if eq else [None] * len(embs_o))
            X = layer.output(y)
            if eq and nxt == 'cross':
                X_eq, X = X, SF.anchor_max(X, dim=0)
        elif block == 'cross':
            if nxt == 'self_eq' or (nxt is None and blocks[i - 1] == 'self_eq'):
                y0 = _cross_plain(layer, P0, P1, X[:R0], X[R0:], X_eq[:, R0:])                      # (A, R0, C)
                x0 = SF.anchor_max(y0, dim=0)
                y1 = _cross_plain(layer, P1, P0, X[R0:], x0, y0)
                X_eq = torch.cat((y0, y1), 1)
                X = torch.cat((x0, SF.anchor_max(y1, dim=0)), 0)
            else:
                x0 = _cross_plain(layer, P0, P1, X[..., :R0, :], X[..., R0:, :], X[..., R0:, :])
                x1 = _cross_plain(layer, P1, P0, X[..., R0:, :], x0, x0)
                X = torch.cat((x0, x1), -2)
        else:
            y0, mixes0 = _cross_eq(layer, P0, P1, X[:, :R0], X[:, R0:])
            y1, _ = _cross_eq(layer, P1, P0, X[:, R0:], y0)
            X = torch.cat((y0, y1), 1)
            if block == 'cross_r_soft' and nxt is not None and not _block_is_eq(nxt):
                # eq2inv_soft: src features re-expressed in the frame the ref<-src rotation weights prefer, per pair
                if y1.shape[0] == 6 and y1.shape[2] % 4 == 0:      # all pairs in one launch (csrc/rowops.hip: anchor_mix_stack_kernel)
                    y1p = _ops.anchor_mix_stack(y1.contiguous(), mixes0, P1.starts, P1.lengths)
                else:
                    y1p = torch.zeros_like(y1)
                    for (s1, n1), mix in zip(zip(P1.starts, P1.lengths), mixes0):
                        y1p[:, s1:s1 + n1] = SF.rotation_weighted_permute(y1[None, :, s1:s1 + n1], mix)[0]
                X = torch.cat((tr.rotcompress(y0[None])[0], tr.rotcompress(y1p[None])[0]), 0)
                X_eq = None
    X = SF.linear(X, gt.out_proj.weight, gt.out_proj.bias)
    if packed:
        return X, PA
    outs = PA.unpack(X)
    return outs[:B], outs[B:]


def registration_pairs(lgr, patches, stacked=None):
    """LocalGlobalRegistration.forward for several pairs at once (same arithmetic per pair; one host synchronisation in all).
    patches[p] = (ref_knn_points (B_p, K, 3), src_knn_points, ref_knn_masks (B_p, K), src_knn_masks, score_mat (B_p, K, K) log).
    Returns per pair (ref_corr_points, src_corr_points, corr_scores, estimated_transform)."""
    P = len(patches)
    nb = [t[0].shape[0] for t in patches]
    dev = patches[0][0].device
    if stacked is not None:                # the caller already holds the pair-major concatenations
        ref_pts, src_pts, ref_masks, src_masks, log_score = stacked
    else:
        ref_pts = torch.cat([t[0] for t in patches], 0)
        src_pts = torch.cat([t[1] for t in patches], 0)
        ref_masks, src_masks = torch.cat([t[2] for t in patches], 0), torch.cat([t[3] for t in patches], 0)
        log_score = torch.cat([t[4] for t in patches], 0)
    score = torch.exp(log_score)
    BT = score.shape[0]
    corr = SF.mutual_topk_mask(score, ref_masks, src_masks, lgr.k, lgr.confidence_threshold)
    b_idx, r_idx, c_idx = torch.nonzero(corr, as_tuple=True)                # the one host sync; patch-major = pair-major
    ref_c, src_c = ref_pts[b_idx, r_idx].contiguous(), src_pts[b_idx, c_idx].contiguous()
    sc = score[b_idx, r_idx, c_idx].contiguous()
    counts = torch.zeros(BT, dtype=torch.int64, device=dev).index_add_(0, b_idx, torch.ones_like(b_idx))   # (bincount syncs)
    offsets = torch.zeros(BT + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(counts, 0)
    patch_off = _offsets(nb)                                                  # host: patches of pair p = [patch_off[p], patch_off[p+1])
    bounds = offsets[_ops.to_device(patch_off, torch.int64, dev)]             # (P + 1,) correspondence range of every pair
    pair_of_patch = _ops.to_device([p for p in range(P) for _ in range(nb[p])], torch.int64, dev)
    # local hypotheses: one weighted Procrustes per patch pair, voted on by the correspondences of ITS pair
    Ts = SF.weighted_procrustes(src_c, ref_c, sc, offsets)
    votes = SF.count_inliers(src_c, ref_c, Ts, lgr.acceptance_radius, bounds[pair_of_patch], bounds[pair_of_patch + 1])
    votes = torch.where(counts >= lgr.correspondence_threshold, votes, torch.full_like(votes, -1))
    if len(set(nb)) == 1:
        v = votes.view(P, nb[0])
        best = torch.argmax(v, 1) + torch.arange(P, device=dev) * nb[0]       # first maximum in patch order, per pair
        any_valid = v.amax(1) >= 0
    else:
        best = torch.stack([torch.argmax(votes[patch_off[p]:patch_off[p + 1]]) + patch_off[p] for p in range(P)])
        any_valid = torch.stack([votes[patch_off[p]:patch_off[p + 1]].max() >= 0 for p in range(P)])
    T_all = SF.weighted_procrustes(src_c, ref_c, sc, bounds)                  # degenerate case: all correspondences of the pair
    T = torch.where(any_valid[:, None, None], Ts[best], T_all)
    for _ in range(lgr.num_refinement_steps):
        T = SF.weighted_procrustes(src_c, ref_c, sc, bounds, gate_transform=T, gate_radius=lgr.acceptance_radius)
    # per-pair slices of the outputs: a second (tiny) sync, except with one pair (all correspondences are its own)
    cuts = [0, int(b_idx.shape[0])] if P == 1 else bounds.tolist()
    return [(ref_c[cuts[p]:cuts[p + 1]], src_c[cuts[p]:cuts[p + 1]], sc[cuts[p]:cuts[p + 1]], T[p]) for p in range(P)]


@torch.no_grad()
def forward_pairs(model, data_dict, with_registration=True):
    """Inference forward of B pairs stacked as ref0, src0, ref1, src1, ... (data_dict from se3et_amd.data with 2 B lengths).
    Returns a list of B output dicts with the keys of SE3ET.forward."""
    lengths = data_dict['lengths']
    nc = len(lengths[0])
    if nc % 2 or nc < 2:
        raise RuntimeError('forward_pairs: an even number of stacked clouds is required (ref0, src0, ref1, src1, ...)')
    B = nc // 2
    # GroupNorm statistics per pair, at every pyramid stage
    seg = []
    for ln in lengths:
        o = _offsets(ln.tolist())
        seg.append([o[2 * p] for p in range(B)] + [o[-1]])
    points_c, points_f = data_dict['points'][-1], data_dict['points'][1]
    len_c, len_f = lengths[-1].tolist(), lengths[1].tolist()
    oc, of = _offsets(len_c), _offsets(len_f)
    points_0, o0 = data_dict['points'][0], _offsets(lengths[0].tolist())
    dev = points_c.device
    K = model.num_points_in_patch
    # every fine point to its nearest superpoint, every superpoint's K nearest own points: all clouds in one call, GLOBAL indices.  It needs
    # the points only, so it goes first: the number of non-empty nodes per cloud travels to the host behind the backbone and the transformer
    # (read after them, when it has long arrived -- no wait, as in SE3ET._forward)
    _, node_masks, knn, knn_masks = _ops.point_to_node_partition_stack(points_f, points_c, len_f, len_c, K)
    csum = torch.cumsum(node_masks, 0)
    ends = _ops.to_device([o - 1 for o in oc[1:]], torch.int64, dev)
    upto = csum[ends]
    valid_host = torch.empty(len(len_c), dtype=torch.int64).pin_memory()
    valid_host.copy_(upto - torch.cat((upto.new_zeros(1), upto[:-1])), non_blocking=True)
    valid_event = torch.cuda.Event()
    valid_event.record()

    if BACKBONE_SPLIT_STAGE is None:
        with SF.norm_segments(seg), _chained('backbone'):
            feats_list = model.backbone(data_dict['features'], data_dict)
    else:
        # SE3_BACKBONE_SPLIT=<stage>: the backbone as TWO kinds of chained sections -- 'backbone_fine' while it works on pyramid stages below
        # <stage> (the long, HBM-heavy kernels), 'backbone' above -- switched where the backbone announces its stage (round 6 experiment:
        # SE3_CHAIN_GROUPS=transformer+backbone_fine keeps only the fine stages off the GPU while another batch's transformer runs)
        import contextlib
        stack = contextlib.ExitStack()
        state = {'name': None}

        def switch(stage):
            name = None if stage is None else ('backbone_fine' if stage < BACKBONE_SPLIT_STAGE else 'backbone')
            if name != state['name']:
                stack.close()
                state['name'] = name
                if name is not None:
                    stack.enter_context(_chained(name))
        with SF.norm_segments(seg):
            SF.norm_segments._tls.stage_hook = switch
            try:
                feats_list = model.backbone(data_dict['features'], data_dict)
            finally:
                SF.norm_segments._tls.stage_hook = None
                stack.close()
    feats_c, feats_f = feats_list[-1], feats_list[0]

    X, PA = transformer_pairs(model.transformer, points_c, len_c, feats_c, packed=True)
    Xn = F.normalize(X, p=2, dim=1)                       # all clouds at once (the padding rows of the packing stay zero)
    valid_event.synchronize()
    valid = valid_host.tolist()                           # non-empty nodes per cloud
    ref_rows, src_rows = PA.starts[:B], PA.starts[B:]
    Ns, Ms = [len_c[2 * p] for p in range(B)], [len_c[2 * p + 1] for p in range(B)]
    ref_off, src_off = [oc[2 * p] for p in range(B)], [oc[2 * p + 1] for p in range(B)]
    cm = model.coarse_matching
    S = _ops.superpoint_scores_stack(Xn, node_masks, ref_rows, src_rows, Ns, Ms, ref_off, src_off, cm.dual_normalization)
    ks = [min(cm.num_correspondences, valid[2 * p] * valid[2 * p + 1]) for p in range(B)]
    if len(set(ks)) == 1:
        node_scores, flat = S.topk(k=ks[0], dim=1, largest=True)             # (B, k): one selection for all pairs
        Mt = _ops.to_device(Ms, torch.int64, dev)[:, None]
        ri = torch.div(flat, Mt, rounding_mode='floor')
        si = flat - ri * Mt
        gr = (ri + _ops.to_device(ref_off, torch.int64, dev)[:, None]).view(-1)      # global superpoint indices, pair-major
        gs = (si + _ops.to_device(src_off, torch.int64, dev)[:, None]).view(-1)
        ri, si, node_scores = list(ri), list(si), list(node_scores)
    else:                                                                     # pairs with fewer valid superpoint pairs than k
        ri, si, node_scores = [], [], []
        for p in range(B):
            v, flat = S[p, :Ns[p] * Ms[p]].topk(k=ks[p], largest=True)
            r = torch.div(flat, Ms[p], rounding_mode='floor')
            ri.append(r)
            si.append(flat - r * Ms[p])
            node_scores.append(v)
        gr = torch.cat([r + o for r, o in zip(ri, ref_off)])
        gs = torch.cat([c + o for c, o in zip(si, src_off)])
    ref_ck, src_ck = knn[gr], knn[gs]                                         # (sum k, K) global fine-point indices
    ref_cm, src_cm = knn_masks[gr], knn_masks[gs]
    ref_cp, src_cp = SF.gather_rows_padded(points_f, ref_ck), SF.gather_rows_padded(points_f, src_ck)
    fused_scores = _ops.patch_scores_ok(feats_f, K)
    if not fused_scores:
        rk, sk = SF.gather_rows_padded(feats_f, ref_ck), SF.gather_rows_padded(feats_f, src_ck)
    outs, patches = [], []
    po = _offsets(ks)
    for p in range(B):
        r, s = 2 * p, 2 * p + 1
        a, b = po[p], po[p + 1]
        outs.append(dict(
            ref_points_c=points_c[oc[r]:oc[r + 1]], src_points_c=points_c[oc[s]:oc[s + 1]],
            ref_points_f=points_f[of[r]:of[r + 1]], src_points_f=points_f[of[s]:of[s + 1]],
            ref_points=points_0[o0[r]:o0[r + 1]], src_points=points_0[o0[s]:o0[s + 1]],
            feats_c=feats_c[oc[r]:oc[s + 1]], feats_f=feats_f[of[r]:of[s + 1]],
            ref_feats_c=Xn[ref_rows[p]:ref_rows[p] + Ns[p]], src_feats_c=Xn[src_rows[p]:src_rows[p] + Ms[p]],
            ref_feats_f=feats_f[of[r]:of[r + 1]], src_feats_f=feats_f[of[s]:of[s + 1]],
            ref_node_corr_indices=ri[p], src_node_corr_indices=si[p], node_corr_scores=node_scores[p],
            ref_node_corr_knn_points=ref_cp[a:b], src_node_corr_knn_points=src_cp[a:b],
            ref_node_corr_knn_masks=ref_cm[a:b], src_node_corr_knn_masks=src_cm[a:b]))
        patches.append((None, None, ref_cm[a:b], src_cm[a:b], ref_cp[a:b], src_cp[a:b], node_scores[p]))
    counts = ks
    # all patch pairs of all registration pairs through ONE Sinkhorn launch
    if fused_scores:      # gathers + products + scale in one kernel (csrc/matching.hip: patch_scores_kernel)
        scores = _ops.patch_scores(feats_f, ref_ck, src_ck, 1.0 / feats_f.shape[1] ** 0.5)
    else:
        scores = torch.einsum('bnd,bmd->bnm', rk, sk) / feats_f.shape[1] ** 0.5
    scores = model.optimal_transport(scores, ref_cm, src_cm)
    start, per_pair = 0, []
    for out, n, t in zip(outs, counts, patches):
        sc = scores[start:start + n]
        start += n
        out['matching_scores'] = sc
        per_pair.append((t[4], t[5], t[2], t[3], sc[:, :-1, :-1]))
    if with_registration:
        stacked = (ref_cp, src_cp, ref_cm, src_cm, scores[:, :-1, :-1])
        for out, (rc, scp, cs, T) in zip(outs, registration_pairs(model.fine_matching, per_pair, stacked)):
            out.update(ref_corr_points=rc, src_corr_points=scp, corr_scores=cs, estimated_transform=T)
    return outs
