"""GPU parity of single HIP ops (through the C ABI) against the oracle on seeded inputs and against per-op
inputs/outputs captured from the reference (tests/golden/micro_se3ete.npz, keys op/*)."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _golden(golden_dir, name='micro_se3ete.npz'):
    return np.load(golden_dir + '/' + name)


def _state(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}


@pytest.mark.parametrize('B,R,C,frac', [(256, 64, 64, 0.7), (37, 64, 64, 1.0), (5, 128, 128, 0.5), (3, 17, 40, 0.8)])
def test_sinkhorn_matches_oracle(B, R, C, frac):
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(B, R, C, generator=g) * 3
    rm, cm = torch.rand(B, R, generator=g) < frac, torch.rand(B, C, generator=g) < frac
    if frac > 0:
        rm[:, 0], cm[:, 0] = True, True
    alpha = torch.tensor(1.3)
    want = O.log_optimal_transport(scores, rm, cm, alpha, 100)
    got = SF.log_optimal_transport(scores.cuda(), rm.cuda(), cm.cuda(), alpha.cuda(), 100, 1e12).cpu()
    valid = want > -1e11
    if frac > 0:
        assert torch.equal(got > -1e11, valid)
        assert float((got[valid] - want[valid]).abs().max()) <= 1e-4 * float(want[valid].abs().max())
        # marginals of the transport plan exp(out) (size independent; the last Sinkhorn half-step fixes the columns exactly, the rows
        # have converged after 100 iterations): every valid point carries mass 1, the dustbins the number of valid points opposite
        plan = torch.exp(got.double()) * valid
        nvr, nvc = rm.sum(1).double(), cm.sum(1).double()
        col, row = plan.sum(1), plan.sum(2)
        assert float((col[:, :-1][cm] - 1).abs().max()) < 1e-4 and float((col[:, -1] - nvr).abs().max() / nvr.max()) < 1e-4
        # rows: as converged as the oracle's own plan after the same 100 iterations (that residual is the algorithm's, ~1e-3 here)
        wrow = (torch.exp(want.double()) * valid).sum(2)
        assert float((row - wrow).abs().max() / nvc.max()) < 1e-4
    assert torch.isfinite(got).all()


@pytest.mark.parametrize('scale', [8.0, 14.0, 30.0])
def test_sinkhorn_wide_score_ranges_match_oracle(scale):
    """Score ranges around and beyond the bound of the scaling form (csrc/sinkhorn.hip: valid scores within 40 of their row's maximum): patches
    on either side of it -- the workgroup-uniform choice between the scaling form and the log-domain loop -- against the oracle."""
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(int(scale))
    B, R, C = 48, 64, 64
    scores = torch.randn(B, R, C, generator=g) * scale
    scores[::2] *= 0.25                                   # every other patch stays narrow
    rm, cm = torch.rand(B, R, generator=g) < 0.8, torch.rand(B, C, generator=g) < 0.8
    rm[:, 0], cm[:, 0] = True, True
    alpha = torch.tensor(0.7)
    want = O.log_optimal_transport(scores, rm, cm, alpha, 100)
    got = SF.log_optimal_transport(scores.cuda(), rm.cuda(), cm.cuda(), alpha.cuda(), 100, 1e12).cpu()
    valid = want > -1e11
    assert torch.equal(got > -1e11, valid)
    assert torch.isfinite(got).all()
    assert float((got[valid] - want[valid]).abs().max()) <= 1e-4 * float(want[valid].abs().max())


@pytest.mark.parametrize('B,R,C,frac,scale,iters', [(64, 64, 64, 0.8, 3.0, 100), (16, 64, 64, 0.05, 3.0, 100), (16, 64, 64, 1.0, 40.0, 100),
                                                    (8, 128, 128, 0.6, 8.0, 100), (8, 17, 40, 0.7, 3.0, 100), (8, 64, 64, 0.8, 3.0, 1),
                                                    (8, 64, 64, 0.8, 3.0, 0), (4, 1, 1, 1.0, 1.0, 100)])
def test_sinkhorn_forms_agree(B, R, C, frac, scale, iters):
    """csrc/sinkhorn.hip: the default loop (base 2, one v_exp_f32 per entry and pass, the previous pass's logsumexp as the shift with the exact
    form as the per-wave fallback) against the reference-order loop kept in the library (se3_debug_set_sinkhorn_variant(1)): nearly empty
    masks, scores 40 wide (every pass of some rows falls back), non-square patches, zero / one iteration, a 1 x 1 patch."""
    from se3et_amd import functional as SF
    from se3et_amd._lib import lib
    g = torch.Generator().manual_seed(B + R + iters)
    scores = (torch.randn(B, R, C, generator=g) * scale).cuda()
    rm, cm = (torch.rand(B, R, generator=g) < frac).cuda(), (torch.rand(B, C, generator=g) < frac).cuda()
    rm[:, 0], cm[:, 0] = True, True
    alpha = torch.tensor(0.9).cuda()
    try:
        lib().se3_debug_set_sinkhorn_variant(1)
        want = SF.log_optimal_transport(scores, rm, cm, alpha, iters, 1e12).cpu()
    finally:
        lib().se3_debug_set_sinkhorn_variant(0)
    got = SF.log_optimal_transport(scores, rm, cm, alpha, iters, 1e12).cpu()
    valid = want > -1e11
    assert torch.equal(got > -1e11, valid) and torch.isfinite(got).all()
    assert float((got[valid] - want[valid]).abs().max()) <= 2e-5 * float(want[valid].abs().max().clamp_min(1.0))


def test_sinkhorn_matches_reference_fixture(golden_dir):
    from se3et_amd import functional as SF
    g = _golden(golden_dir)
    scores = torch.from_numpy(g['op/sinkhorn/in0'])
    rm, cm = torch.from_numpy(g['op/sinkhorn/in1']), torch.from_numpy(g['op/sinkhorn/in2'])
    want = torch.from_numpy(g['op/sinkhorn/out0'])
    alpha = torch.from_numpy(g['sd/optimal_transport.alpha'])
    got = SF.log_optimal_transport(scores.cuda(), rm.cuda(), cm.cuda(), alpha.cuda(), 100, 1e12).cpu()
    valid = want > -1e11
    assert torch.equal(got > -1e11, valid)
    assert float((got[valid] - want[valid]).abs().max()) < 1e-4 * float(want[valid].abs().max())


@pytest.mark.parametrize('rows,A,C,G', [(1000, 6, 64, 32), (5003, 6, 16, 4), (700, 6, 256, 32), (333, 1, 512, 32),
                                        (10000, 6, 1, 1), (64, 6, 1024, 32)])
def test_group_norm_matches_oracle(rows, A, C, G):
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(1)
    shape = (rows, A, C) if A > 1 else (rows, C)
    x = torch.randn(shape, generator=g) * 2 + 3.0           # non-zero mean: exercises the cancellation-free statistics
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    res = torch.randn(shape, generator=g)
    want = (O.group_norm_epn(x, w, b, G) if A > 1 else O.group_norm_flat(x, w, b, G))
    got = SF.group_norm_rows(x.cuda(), w.cuda(), b.cuda(), G, 1e-5, None, None).cpu()
    assert_close(got, want, 1e-5, 'group norm')
    got2 = SF.group_norm_rows(x.cuda(), w.cuda(), b.cuda(), G, 1e-5, 0.1, res.cuda()).cpu()
    assert_close(got2, torch.nn.functional.leaky_relu(want + res, 0.1), 1e-5, 'group norm + residual + lrelu')
    # bias of the producing linear folded into the kernels: GN(x + xb) with x passed bias-free
    xb = torch.randn(C, generator=g) * 3
    want3 = (O.group_norm_epn(x + xb, w, b, G) if A > 1 else O.group_norm_flat(x + xb, w, b, G))
    got3 = SF.group_norm_rows(x.cuda(), w.cuda(), b.cuda(), G, 1e-5, 0.1, res.cuda(), x_bias=xb.cuda()).cpu()
    assert_close(got3, torch.nn.functional.leaky_relu(want3 + res, 0.1), 1e-5, 'group norm with folded input bias')
    # segments: independent statistics per row range (several pairs in one launch) == one call per range
    if rows >= 64:
        cuts = [0, rows // 3, rows // 3 + 7, rows]
        seg_rows = [c * (A if A > 1 else 1) for c in cuts]
        got4 = SF.group_norm_rows(x.cuda(), w.cuda(), b.cuda(), G, 1e-5, 0.1, res.cuda(), x_bias=xb.cuda(), segments=seg_rows).cpu()
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            xs = x[lo:hi] + xb
            ws_ = (O.group_norm_epn(xs, w, b, G) if A > 1 else O.group_norm_flat(xs, w, b, G))
            assert_close(got4[lo:hi], torch.nn.functional.leaky_relu(ws_ + res[lo:hi], 0.1), 1e-5, 'segmented group norm')


@pytest.mark.parametrize('A,N,C', [(6, 382, 256), (1, 59, 32), (6, 53, 128), (1, 300, 1024)])
def test_add_layer_norm_matches_torch(A, N, C):
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(2)
    h = torch.randn(1, A, N, C, generator=g) if A > 1 else torch.randn(1, N, C, generator=g)
    r = torch.randn(1, N, C, generator=g)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rr = r.unsqueeze(1) if A > 1 else r
    want = torch.nn.functional.layer_norm(h + rr, (C,), w, b, 1e-5)
    got = SF.add_layer_norm(h.cuda(), rr.cuda(), w.cuda(), b.cuda(), 1e-5).cpu()
    assert_close(got, want, 1e-5, 'add+LN')
    hb = torch.randn(C, generator=g) * 2
    want2 = torch.nn.functional.layer_norm(h + hb + rr, (C,), w, b, 1e-5)
    got2 = SF.add_layer_norm(h.cuda(), rr.cuda(), w.cuda(), b.cuda(), 1e-5, hidden_bias=hb.cuda()).cpu()
    assert_close(got2, want2, 1e-5, 'add+LN with folded linear bias')


@pytest.mark.parametrize('shape,dim', [((6, 5760, 256), 0), ((5003, 6, 64), 1), ((6, 37, 32), 0), ((1, 6, 4), 1)])
def test_anchor_max_equals_torch(shape, dim):
    from se3et_amd import functional as SF
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3)).cuda()
    assert torch.equal(SF.anchor_max(x, dim=dim), x.amax(dim))
    wide = torch.randn(shape[0], shape[1], 2 * shape[2], generator=torch.Generator().manual_seed(4)).cuda()
    view = wide[:, :, shape[2]:]                      # strided view (column block of a wider tensor)
    assert torch.equal(SF.anchor_max(view, dim=dim), view.amax(dim))


def test_gather_and_max_pool_match_oracle():
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(3)
    x = torch.randn(500, 6, 16, generator=g)
    idx = torch.randint(0, 501, (300, 19), generator=g)
    assert torch.equal(SF.neighbor_max_pool(x.cuda(), idx.cuda()).cpu(), O.max_pool(x, idx))
    want = torch.cat((x, torch.zeros_like(x[:1])), 0)[idx[:, 0]]
    assert torch.equal(SF.gather_rows_padded(x.cuda(), idx[:, 0].contiguous().cuda()).cpu(), want)
    want2 = torch.cat((x, torch.zeros_like(x[:1])), 0)[idx]
    assert torch.equal(SF.gather_rows_padded(x.cuda(), idx.cuda()).cpu(), want2)


def _conv_state(Cin, Cout, radius, seed=0):
    from se3et_amd import tables
    g = torch.Generator().manual_seed(seed)
    return {'kernel_points': torch.from_numpy(tables.kernel_points(radius)),
            'weights': torch.randn(6, 6, Cin, Cout, generator=g) / (36 * Cin) ** 0.5,
            'kidx_rot': torch.from_numpy(tables.kernel_slot_table())[:, None, :].expand(-1, 6, -1).contiguous(),
            'ridx_rot': torch.from_numpy(tables.anchor_slot_table())[None].expand(15, -1, -1).contiguous()}


@pytest.mark.parametrize('P,Ns,NN,Cin,Cout', [(700, 900, 19, 1, 16), (500, 500, 35, 32, 32), (300, 800, 36, 64, 64),
                                              (120, 120, 38, 256, 256), (257, 300, 13, 8, 8), (1000, 1000, 36, 128, 128),
                                              (333, 400, 20, 16, 16), (50, 64, 38, 512, 512), (37, 64, 12, 64, 32)])
def test_kpconv_matches_oracle(P, Ns, NN, Cin, Cout):
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    from se3et_amd import ops
    g = torch.Generator().manual_seed(4)
    radius, sigma = 0.0625, 0.05
    s_pts = torch.rand(Ns, 3, generator=g) * 0.3
    q_pts = s_pts[torch.randperm(Ns, generator=g)[:P]].contiguous() if P <= Ns else torch.rand(P, 3, generator=g) * 0.3
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(NN, dim=1, largest=False)[1]
    idx[d.gather(1, idx) > radius ** 2] = Ns                      # shadow neighbours
    x = torch.randn(Ns, 6, Cin, generator=g)
    st = _conv_state(Cin, Cout, radius)
    want = O.kpconv_inter_so3({'c.' + k: v for k, v in st.items()}, 'c.', q_pts, s_pts, idx, x, sigma)
    saved = ops.KPCONV_MATRIX_CORE
    try:
        for path in (True, False):        # matrix-core kernels (where the channel counts allow) and the gather + library GEMM path
            ops.KPCONV_MATRIX_CORE = path
            got = SF.kpconv_inter_so3(x.cuda(), q_pts.cuda(), s_pts.cuda(), idx.cuda(), st['kernel_points'].cuda(),
                                      st['weights'].cuda(), st['kidx_rot'][:, 0, :].cuda(), st['ridx_rot'][0].cuda(), sigma).cpu()
            assert_close(got, want, 1e-4, 'kpconv (matrix core %s)' % path)
    finally:
        ops.KPCONV_MATRIX_CORE = saved


@pytest.mark.parametrize('P,Ns,NN,Cin,Cout', [(500, 500, 35, 32, 32), (2001, 2500, 36, 128, 128), (129, 200, 38, 256, 256), (1, 50, 20, 64, 128),
                                              (17, 60, 22, 128, 64), (33, 70, 38, 64, 192), (16, 40, 9, 8, 16), (250, 300, 30, 136, 48)])
def test_kpconv_matrix_core_path_has_f32_accuracy(P, Ns, NN, Cin, Cout):
    """The bf16 matrix-core contraction (three-way operand splits, six products: csrc/kpconv_contract.hip) against the f32 path
    (slot sums + library f32 GEMM) on the same inputs: agreement at f32 round-off level (1e-5 of the max), and both against a float64
    evaluation of the same formula, where the split path must not be worse than 2x the f32 path's own error."""
    from se3et_amd import functional as SF
    from se3et_amd import ops
    g = torch.Generator().manual_seed(14)
    radius, sigma = 0.0625, 0.05
    s_pts = torch.rand(Ns, 3, generator=g) * 0.3
    q_pts = s_pts[torch.randperm(Ns, generator=g)[:P]].contiguous()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(NN, dim=1, largest=False)[1]
    idx[d.gather(1, idx) > radius ** 2] = Ns
    x = torch.randn(Ns, 6, Cin, generator=g)
    st = _conv_state(Cin, Cout, radius)
    args = (x.cuda(), q_pts.cuda(), s_pts.cuda(), idx.cuda(), st['kernel_points'].cuda(), st['weights'].cuda(),
            st['kidx_rot'][:, 0, :].cuda(), st['ridx_rot'][0].cuda(), sigma)
    saved = ops.KPCONV_MATRIX_CORE
    try:
        ops.KPCONV_MATRIX_CORE = True
        new = SF.kpconv_inter_so3(*args).cpu()
        ops.KPCONV_MATRIX_CORE = False
        old = SF.kpconv_inter_so3(*args).cpu()
    finally:
        ops.KPCONV_MATRIX_CORE = saved
    assert_close(new, old, 1e-5, 'matrix-core path vs f32 path')
    # float64 evaluation of out = sum_{k,a,c} F[k,a,c] W[kidx[k,r], ridx[a,r], c, d]
    xs = torch.cat((x, torch.zeros(1, 6, Cin))).double()
    sp = torch.cat((s_pts, torch.full((1, 3), 1e6))).double()
    nb = sp[idx] - q_pts.double()[:, None]
    w = (1 - (nb[:, :, None] - st['kernel_points'].double()[None, None]).norm(dim=-1) / sigma).clamp(min=0)      # (P, NN, K)
    Fk = torch.einsum('pnk,pnac->pkac', w, xs[idx])
    W = st['weights'].double()[st['kidx_rot'][:, 0, :][:, None, :], st['ridx_rot'][0][None, :, :]]                # (K, A, R, Cin, Cout)
    ref = torch.einsum('pkac,karcd->prd', Fk, W)
    e_new, e_old = float((new.double() - ref).abs().max()), float((old.double() - ref).abs().max())
    assert e_new <= max(2 * e_old, 2e-6 * float(ref.abs().max())), (e_new, e_old)


def test_kpconv_matches_reference_fixture(golden_dir):
    from se3et_amd import functional as SF
    g = _golden(golden_dir)
    sd = _state(g)
    for name, blk, sigma in (('kpconv_2_2', 'encoder2_2', 0.1), ('kpconv_2_1', 'encoder2_1', 0.05)):
        q, s, idx, x = [torch.from_numpy(g['op/%s/in%d' % (name, i)]) for i in range(4)]
        pre = 'backbone.%s.interso3.conv.' % blk
        got = SF.kpconv_inter_so3(x.cuda(), q.cuda(), s.cuda(), idx.long().cuda(), sd[pre + 'kernel_points'].cuda(),
                                  sd[pre + 'weights'].cuda(), sd[pre + 'kidx_rot'][:, 0, :].cuda(),
                                  sd[pre + 'ridx_rot'][0].cuda(), sigma).cpu()
        assert_close(got, g['op/%s/out0' % name], 1e-4, name)


def _attn_state(C, eq, seed=0):
    g = torch.Generator().manual_seed(seed)
    st = {}
    for n in ('q', 'k', 'v', 'p'):
        st['l.proj_%s.weight' % n] = torch.randn(C, C, generator=g) / C ** 0.5
        st['l.proj_%s.bias' % n] = torch.randn(C, generator=g) * 0.1
    if eq:
        st['l.proj_eq.weight'] = torch.randn(C, 4, generator=g) * 0.5
        st['l.proj_eq.bias'] = torch.randn(C, generator=g) * 0.1
    return st


@pytest.mark.parametrize('A,N,C,H,eq', [(6, 59, 32, 4, True), (6, 382, 256, 4, True), (1, 304, 256, 4, False),
                                        (6, 100, 128, 4, True), (1, 33, 128, 4, False), (6, 17, 64, 4, True)])
def test_rpe_attention_matches_oracle(A, N, C, H, eq):
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(5)
    st = _attn_state(C, eq)
    x = torch.randn((A, N, C) if A > 1 else (N, C), generator=g)
    emb = torch.randn(N, N, C, generator=g)
    eq_emb = torch.randn(A, N, N, 4, generator=g) if eq else None
    want, want_scores = O.rpe_attention(st, 'l.', x, x, emb, eq_emb, H)
    lin = lambda n: torch.nn.functional.linear(x, st['l.proj_%s.weight' % n], st['l.proj_%s.bias' % n]).cuda()
    vt = SF.project_values_transposed(x.cuda(), st['l.proj_v.weight'].cuda(), st['l.proj_v.bias'].cuda())
    got, scores = SF.rpe_attention(lin('q'), lin('k'), vt, emb.cuda(), st['l.proj_p.weight'].cuda(),
                                   eq_emb.cuda() if eq else None, st['l.proj_eq.weight'].cuda() if eq else None, H,
                                   return_scores=True)
    assert_close(got.cpu(), want, 1e-4, 'rpe attention hidden')
    assert_close(scores.cpu(), want_scores, 1e-4, 'rpe attention scores')


@pytest.mark.parametrize('A,lengths,C,H,eq', [(6, (59, 53), 32, 4, True), (6, (382, 350), 256, 4, True), (1, (304, 382), 256, 4, False),
                                              (6, (100,), 128, 4, True), (1, (33, 17, 64), 128, 4, False), (3, (17, 40), 64, 2, True),
                                              (6, (40, 31, 65, 32), 64, 4, True), (6, (410, 300), 256, 4, True), (1, (410, 300), 256, 4, False),
                                              (6, (513,), 256, 4, True)])
def test_rpe_self_attention_stack_matches_oracle(A, lengths, C, H, eq):
    """Stack mode (all clouds of a pair in ONE launch per kernel, se3_rpe_bias_stack_fwd + se3_attention_stack_fwd) against the
    oracle run cloud by cloud; also pins the composed [q | k | W_p^T q | W_eq^T q] projection."""
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(11)
    st = _attn_state(C, eq)
    xs = [torch.randn((A, n, C) if A > 1 else (n, C), generator=g) for n in lengths]
    embs = [torch.randn(n, n, C, generator=g) for n in lengths]
    eqs = [torch.randn(A, n, n, 4, generator=g) if eq else None for n in lengths]
    w_stack, b_stack, offs = SF.compose_self_attention_weights(st['l.proj_q.weight'], st['l.proj_q.bias'], st['l.proj_k.weight'],
                                                               st['l.proj_k.bias'], st['l.proj_p.weight'],
                                                               st['l.proj_eq.weight'] if eq else None, H)
    packed, starts = SF.pack_rows([x.cuda() for x in xs])
    got = SF.rpe_self_attention_packed(packed, starts, list(lengths), [e.cuda() for e in embs],
                                       [e.cuda() if e is not None else None for e in eqs], w_stack.cuda(), b_stack.cuda(), offs,
                                       st['l.proj_v.weight'].cuda(), st['l.proj_v.bias'].cuda(), H).cpu()
    for x, emb, e, s0, n in zip(xs, embs, eqs, starts, lengths):
        want, _ = O.rpe_attention(st, 'l.', x, x, emb, e, H)
        assert_close(got[..., s0:s0 + n, :], want, 1e-4, 'stack-mode rpe attention, cloud at row %d' % s0)
    pad = torch.ones(got.shape[-2], dtype=torch.bool)
    for s0, n in zip(starts, lengths):
        pad[s0:s0 + n] = False
    assert float(got[..., pad, :].abs().max()) == 0.0 if pad.any() else True


@pytest.mark.parametrize('A,lengths,C,H,eq', [(6, (59, 53), 32, 4, True), (6, (382, 350), 256, 4, True), (1, (304, 382), 256, 4, False),
                                              (1, (33, 17, 64), 128, 4, False), (3, (17, 40), 64, 2, True)])
def test_rpe_self_attention_stack_bf16_embedding_matches_oracle(A, lengths, C, H, eq):
    """'bf16 attention' (BASELINE.json configs[2]): the geometric embedding is STORED in bf16 and read by
    se3_rpe_self_attention_stack_bf16_fwd; everything else stays f32 (the folded queries are split hi + lo inside the kernel).
    Against the oracle on the same rounded embedding the f32 tolerance holds."""
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(12)
    st = _attn_state(C, eq)
    xs = [torch.randn((A, n, C) if A > 1 else (n, C), generator=g) for n in lengths]
    embs = [torch.randn(n, n, C, generator=g).to(torch.bfloat16) for n in lengths]
    eqs = [torch.randn(A, n, n, 4, generator=g) if eq else None for n in lengths]
    w_stack, b_stack, offs = SF.compose_self_attention_weights(st['l.proj_q.weight'], st['l.proj_q.bias'], st['l.proj_k.weight'],
                                                               st['l.proj_k.bias'], st['l.proj_p.weight'],
                                                               st['l.proj_eq.weight'] if eq else None, H)
    packed, starts = SF.pack_rows([x.cuda() for x in xs])
    got = SF.rpe_self_attention_packed(packed, starts, list(lengths), [e.cuda() for e in embs],
                                       [e.cuda() if e is not None else None for e in eqs], w_stack.cuda(), b_stack.cuda(), offs,
                                       st['l.proj_v.weight'].cuda(), st['l.proj_v.bias'].cuda(), H).cpu()
    for x, emb, e, s0, n in zip(xs, embs, eqs, starts, lengths):
        want, _ = O.rpe_attention(st, 'l.', x, x, emb.float(), e, H)
        assert_close(got[..., s0:s0 + n, :], want, 1e-4, 'stack-mode rpe attention with a bf16 embedding, cloud at row %d' % s0)


def test_stack_mode_rejects_bad_descriptors():
    from se3et_amd import ops
    q = torch.zeros(1, 64, 32, device='cuda')
    vt = torch.zeros(1, 32, 64, device='cuda')
    out = torch.zeros(1, 64, 32, device='cuda')
    with pytest.raises(RuntimeError):          # more clouds than one launch takes
        ops.attention_stack(q, q, vt, None, None, [0] * 33, [8] * 33, [0] * 33, [8] * 33, 4, out)      # (SE3_MAX_BATCH = 32 clouds)
    with pytest.raises(RuntimeError):          # cloud beyond the packed rows
        ops.attention_stack(q, q, vt, None, None, [32], [40], [32], [40], 4, out)
    with pytest.raises(RuntimeError):          # key columns must start at a multiple of 4
        ops.attention_stack(q, q, vt, None, None, [0], [8], [2], [8], 4, out)


def test_rpe_attention_matches_reference_fixture(golden_dir):
    from se3et_amd import functional as SF
    g = _golden(golden_dir)
    sd = _state(g)
    emb, eq_emb = torch.from_numpy(g['op/embedding/out0'])[0], torch.from_numpy(g['op/embedding/out1'])[0]
    for layer, eq in ((0, True), (4, False)):
        pre = 'transformer.transformer.layers.%d.attention.attention.' % layer
        x = torch.from_numpy(g['op/attn_%d/in0' % layer])[0]
        lin = lambda n: torch.nn.functional.linear(x, sd[pre + 'proj_%s.weight' % n], sd[pre + 'proj_%s.bias' % n]).cuda()
        vt = SF.project_values_transposed(x.cuda(), sd[pre + 'proj_v.weight'].cuda(), sd[pre + 'proj_v.bias'].cuda())
        got, scores = SF.rpe_attention(lin('q'), lin('k'), vt, emb.cuda(), sd[pre + 'proj_p.weight'].cuda(),
                                       eq_emb.cuda() if eq else None, sd[pre + 'proj_eq.weight'].cuda() if eq else None, 4,
                                       return_scores=True)
        assert_close(got.cpu(), g['op/attn_%d/out0' % layer][0], 1e-4, 'layer %d hidden' % layer)
        assert_close(scores.cpu(), g['op/attn_%d/out1' % layer][0], 1e-4, 'layer %d scores' % layer)


@pytest.mark.parametrize('A,N,M,C,H', [(1, 59, 53, 32, 4), (6, 382, 304, 256, 4), (6, 304, 382, 256, 4), (1, 40, 700, 128, 4)])
def test_cross_attention_matches_oracle(A, N, M, C, H):
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(6)
    q, k = torch.randn(N, C, generator=g), torch.randn(M, C, generator=g)
    v = torch.randn((A, M, C) if A > 1 else (M, C), generator=g)
    hs = lambda t: t.reshape(*t.shape[:-1], H, -1).transpose(-2, -3)
    p = torch.softmax(hs(q) @ hs(k).transpose(-1, -2) / (C // H) ** 0.5, -1)
    want = (p @ hs(v)).transpose(-2, -3)
    want = want.reshape(*want.shape[:-2], -1)
    eye, zero = torch.eye(C).cuda(), torch.zeros(C).cuda()
    got = SF.cross_attention(q.cuda(), k.cuda(), SF.project_values_transposed(v.cuda(), eye, zero), H).cpu()
    assert_close(got, want, 1e-4, 'cross attention')


@pytest.mark.parametrize('N,M,C,mode', [(59, 53, 32, 'a_soft'), (59, 53, 32, 'r_soft'), (382, 304, 256, 'a_soft'),
                                        (304, 382, 256, 'r_soft'), (100, 37, 128, 'r_soft')])
def test_cross_attention_eq_matches_oracle(N, M, C, mode):
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    from se3et_amd import tables
    g = torch.Generator().manual_seed(7)
    H = 4
    trace = torch.from_numpy(tables.trace_indices()[0])
    q, k, v = torch.randn(6, N, C, generator=g), torch.randn(6, M, C, generator=g), torch.randn(6, M, C, generator=g)
    hs = lambda t: t.reshape(*t.shape[:-1], H, -1).transpose(-2, -3)
    qh, kh, vh = hs(q), hs(k), hs(v)
    s = torch.einsum('ahnc,ehmc->aehnm', qh, kh) / (C // H) ** 0.5
    gg = (s.mean(2) ** 2).mean((-2, -1))
    p = torch.softmax(s, -1)
    if mode == 'a_soft':
        w = gg / gg.sum(1, keepdim=True)
        hidden = torch.einsum('aehnm,ehmc->ahnc', p * w[:, :, None, None, None], vh)
        want_w = w
    else:
        ar = torch.arange(6)
        w = gg[ar[None, :], trace].mean(1)
        w = w / w.sum()
        hidden = torch.zeros_like(qh)
        for r in range(24):
            hidden = hidden + w[r] * torch.einsum('ahnm,ahmc->ahnc', p[ar, trace[r]], vh[trace[r]])
        want_w = w
    want = hidden.transpose(-2, -3).reshape(6, N, C)
    eye, zero = torch.eye(C).cuda(), torch.zeros(C).cuda()
    got, got_w, got_mix = SF.cross_attention_eq(q.cuda(), k.cuda(), SF.project_values_transposed(v.cuda(), eye, zero), H, mode,
                                       trace.cuda())
    assert_close(got_w.cpu(), want_w, 1e-4, 'global weights')
    assert_close(got.cpu(), want, 1e-4, 'eq cross attention')


@pytest.mark.parametrize('N,C,eq', [(59, 32, True), (382, 256, True), (304, 256, False), (100, 128, True), (4, 64, True), (5, 256, False),
                                     (17, 96, True), (65, 256, True), (531, 64, False), (33, 48, True)])
def test_geometric_embedding_matches_oracle(N, C, eq):
    """Both forms of the kernel: channel slices with the angle table in LDS (C a multiple of 32: tiny clouds leave most waves without a
    pair, 531 points give uneven row blocks) and the single-kernel form (C = 48)."""
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    from se3et_amd import tables
    g = torch.Generator().manual_seed(8)
    pts = torch.rand(N, 3, generator=g) * torch.tensor([1.5, 1.2, 1.0])
    div = torch.exp(torch.arange(0, C, 2).float() * (-np.log(10000.0) / C))
    st = {'e.embedding.div_term': div}
    for n in ('d', 'a'):
        st['e.proj_%s.weight' % n] = torch.randn(C, C, generator=g) / C ** 0.5
        st['e.proj_%s.bias' % n] = torch.randn(C, generator=g) * 0.1
    w0, w1 = [torch.from_numpy(t) for t in tables.wigner_tables()]
    st['e.anchors_wignerD.0'], st['e.anchors_wignerD.1'] = w0, w1
    cfg = O.OracleConfig()
    want = O.geometric_embedding(st, 'e.', pts, cfg)
    c = lambda k: st[k].cuda()
    out = SF.geometric_embedding(pts.cuda(), c('e.embedding.div_term'), c('e.proj_d.weight'), c('e.proj_d.bias'),
                                 c('e.proj_a.weight'), c('e.proj_a.bias'), cfg.sigma_d, cfg.sigma_a, 3,
                                 wigner_d1=w1.cuda() if eq else None)
    emb = (out[0] if eq else out).cpu()
    # the n == m diagonal of the distance index is sqrt of the rounding residue of |x|^2 - 2 x.y + |y|^2 in the reference
    # (pure noise, ~1e-3 index units, BLAS dependent); everything else must agree to 1e-4
    off = ~torch.eye(N, dtype=torch.bool)
    assert_close(emb[off], want[off], 1e-4, 'geometric embedding (off-diagonal)')
    assert_close(emb, want, 2e-3, 'geometric embedding (diagonal noise)')
    if eq:
        assert_close(out[1].cpu(), O.equiv_embedding(st, 'e.', pts), 1e-5, 'equivariant embedding')


def test_geometric_embedding_bf16_is_the_rounded_f32_embedding():
    """se3_geo_embedding_bf16_fwd stores round-to-nearest-even(bf16) of exactly the value the f32 entry writes."""
    from se3et_amd import functional as SF
    from se3et_amd import tables
    g = torch.Generator().manual_seed(9)
    N, C = 211, 256
    pts = (torch.rand(N, 3, generator=g) * torch.tensor([1.5, 1.2, 1.0])).cuda()
    div = torch.exp(torch.arange(0, C, 2).float() * (-np.log(10000.0) / C)).cuda()
    w = [(torch.randn(C, C, generator=g) / C ** 0.5).cuda() for _ in range(2)]
    b = [(torch.randn(C, generator=g) * 0.1).cuda() for _ in range(2)]
    w1 = torch.from_numpy(tables.wigner_tables()[1]).cuda()
    e32, q32 = SF.geometric_embedding(pts, div, w[0], b[0], w[1], b[1], 0.2, 15.0, 3, wigner_d1=w1)
    e16, q16 = SF.geometric_embedding(pts, div, w[0], b[0], w[1], b[1], 0.2, 15.0, 3, wigner_d1=w1, dtype=torch.bfloat16)
    assert e16.dtype == torch.bfloat16 and q16.dtype == torch.float32
    assert torch.equal(e16, e32.to(torch.bfloat16))
    assert torch.equal(q16, q32)


def test_geometric_embedding_out_of_table_range():
    """Distance indices beyond the tabulated range (here sigma_d = 0.002 -> indices up to ~800) take the exact-sum path."""
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(9)
    N, C = 40, 32
    pts = torch.rand(N, 3, generator=g)
    div = torch.exp(torch.arange(0, C, 2).float() * (-np.log(10000.0) / C))
    st = {'e.embedding.div_term': div}
    for n in ('d', 'a'):
        st['e.proj_%s.weight' % n] = torch.randn(C, C, generator=g) / C ** 0.5
        st['e.proj_%s.bias' % n] = torch.randn(C, generator=g) * 0.1
    cfg = O.OracleConfig(sigma_d=0.002)
    want = O.geometric_embedding(st, 'e.', pts.double(), cfg) if False else O.geometric_embedding(st, 'e.', pts, cfg)
    c = lambda k: st[k].cuda()
    got = SF.geometric_embedding(pts.cuda(), c('e.embedding.div_term'), c('e.proj_d.weight'), c('e.proj_d.bias'),
                                 c('e.proj_a.weight'), c('e.proj_a.bias'), cfg.sigma_d, cfg.sigma_a, 3).cpu()
    off = ~torch.eye(N, dtype=torch.bool)
    # the index x = d / sigma_d amplifies the fp32 rounding of d (both sides) by 1 / sigma_d: 1e-3 is the noise floor here
    assert_close(got[off], want[off], 2e-3, 'embedding, exact path')


def test_geometric_embedding_follows_in_place_weight_updates():
    """The tabulated W emb(x) + b is validated against the CURRENT weight values on the device before every use: writes that
    keep the Parameter's identity, version counter and address (`p.data.copy_`), a module moved to another dtype and back, and
    load_state_dict must all be reflected by the next forward (a host-side cache keyed on id/_version went stale)."""
    from oracle import se3et_oracle as O
    from se3et_amd.modules.geotransformer.geotransformer import GeometricStructureEmbedding
    g = torch.Generator().manual_seed(10)
    N, C = 97, 64
    pts = torch.rand(N, 3, generator=g)
    mod = GeometricStructureEmbedding(C, 0.2, 15, 3, kanchor=6, n_level_equiv=0).cuda()
    cfg = O.OracleConfig()
    off = ~torch.eye(N, dtype=torch.bool)

    def want():
        st = {'e.' + k: v.detach().cpu() for k, v in mod.state_dict().items()}
        return O.geometric_embedding(st, 'e.', pts, cfg)

    first = mod(pts.cuda().unsqueeze(0))[0].cpu()
    assert_close(first[off], want()[off], 1e-4, 'initial weights')
    version = mod.proj_d.weight._version
    mod.proj_d.weight.data.copy_(torch.randn(C, C, generator=g) / C ** 0.5)          # same id, same _version, same address
    assert mod.proj_d.weight._version == version
    second = mod(pts.cuda().unsqueeze(0))[0].cpu()
    assert float((second - first).abs().max()) > 1e-2
    assert_close(second[off], want()[off], 1e-4, 'after proj_d.weight.data.copy_')
    mod.proj_a.bias.data.add_(0.25)
    third = mod(pts.cuda().unsqueeze(0))[0].cpu()
    assert_close(third[off], want()[off], 1e-4, 'after proj_a.bias.data.add_')
    sd = {k: (torch.randn(v.shape, generator=g) * 0.1 if k.startswith('proj_') else v.cpu()) for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd)
    mod = mod.cpu().cuda()                                                            # module.to(): new storage, same Parameters
    fourth = mod(pts.cuda().unsqueeze(0))[0].cpu()
    assert_close(fourth[off], want()[off], 1e-4, 'after load_state_dict + device round trip')


def test_geometric_embedding_matches_reference_fixture(golden_dir):
    from se3et_amd import functional as SF
    g = _golden(golden_dir)
    sd = _state(g)
    pts = torch.from_numpy(g['op/embedding/in0'])[0]
    p = 'transformer.embedding.'
    c = lambda k: sd[p + k].cuda()
    emb, eq = SF.geometric_embedding(pts.cuda(), c('embedding.div_term'), c('proj_d.weight'), c('proj_d.bias'),
                                     c('proj_a.weight'), c('proj_a.bias'), 0.2, 15, 3, wigner_d1=c('anchors_wignerD.1'))
    off = ~torch.eye(pts.shape[0], dtype=torch.bool)
    assert_close(emb.cpu()[off], torch.from_numpy(g['op/embedding/out0'][0])[off], 1e-4, 'embedding vs reference')
    assert_close(emb.cpu(), g['op/embedding/out0'][0], 2e-3, 'embedding vs reference (diagonal noise)')
    assert_close(eq.cpu(), g['op/embedding/out1'][0], 1e-5, 'eq embedding vs reference')


def test_pairwise_distance_matches_reference_formula():
    """E1 (modules/ops/pairwise_distance.py:4-30): x2 - 2xy + y2 (or 2 - 2xy for unit vectors), clamped at 0, channel-first option."""
    from se3et_amd.modules.ops import pairwise_distance
    g = torch.Generator().manual_seed(32)
    x, y = torch.randn(3, 41, 16, generator=g), torch.randn(3, 29, 16, generator=g)
    want = ((x[:, :, None] - y[:, None]) ** 2).sum(-1)
    assert_close(pairwise_distance(x.cuda(), y.cuda()).cpu(), want, 1e-5, 'pairwise distance')
    assert_close(pairwise_distance(x.transpose(1, 2).contiguous().cuda(), y.transpose(1, 2).contiguous().cuda(), channel_first=True).cpu(), want, 1e-5,
                 'pairwise distance, channel first')
    xn, yn = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(y, dim=-1)
    assert_close(pairwise_distance(xn.cuda(), yn.cuda(), normalized=True).cpu(), ((xn[:, :, None] - yn[:, None]) ** 2).sum(-1), 1e-5, 'normalized')
    assert float(pairwise_distance(xn.cuda(), xn.cuda(), normalized=True).min()) >= 0.0          # clamped: no negative round-off on the diagonal


@pytest.mark.parametrize('shape', [((), 382, 350, 256), ((), 5000, 382, 3), ((2,), 65, 64, 3), ((3, 2), 17, 130, 20), ((), 1, 1, 1),
                                   ((4,), 128, 65, 7), ((), 0, 5, 3), ((), 700, 1, 32)])
@pytest.mark.parametrize('normalized', [False, True])
def test_pairwise_distance_kernel_matches_oracle(shape, normalized):
    """csrc/pairwise_distance.hip against the oracle's restatement of pairwise_distance.py:4-30 (which is pinned to the reference by the
    model fixtures) and against a float64 evaluation: superpoint features (C = 256), point coordinates (C = 3: the scalar-load
    instantiation), batch dimensions, sizes that are not multiples of the 64 x 64 block, empty and single-row inputs; the gradient path
    stays on the reference formula."""
    from oracle import se3et_oracle as O
    from se3et_amd import ops
    from se3et_amd.modules.ops import pairwise_distance
    batch, N, M, C = shape
    g = torch.Generator().manual_seed(N * 7 + M + C)
    x, y = torch.randn(batch + (N, C), generator=g), torch.randn(batch + (M, C), generator=g)
    if normalized:
        x, y = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(y, dim=-1)
    got = pairwise_distance(x.cuda(), y.cuda(), normalized=normalized)
    assert tuple(got.shape) == batch + (N, M)
    if N == 0:
        return
    want = O.pairwise_distance(x, y, normalized=normalized)
    ref = ((x.double()[..., :, None, :] - y.double()[..., None, :, :]) ** 2).sum(-1)
    scale = float(ref.abs().max())
    assert float((got.cpu() - want).abs().max()) <= 1e-5 * scale
    # not further from the exact distances than the reference formula in float32 (plus a rounding of the largest term)
    e_new, e_ref = float((got.cpu().double() - ref).abs().max()), float((want.double() - ref).abs().max())
    assert e_new <= 2 * e_ref + 4e-7 * scale, (e_new, e_ref)
    assert float(got.min()) >= 0.0
    if N > 1 and not batch:
        torch.cuda.synchronize()
        assert torch.equal(ops.pairwise_distance(x.cuda(), y.cuda(), normalized), got)                       # same launch, same bits
        xg = x.cuda().requires_grad_(True)                                                                   # losses differentiate through it
        d = pairwise_distance(xg, y.cuda(), normalized=normalized)
        d.sum().backward()
        assert xg.grad is not None and float((d.detach() - got).abs().max()) <= 1e-5 * scale


def test_key_masks_take_the_minus_infinity_path():
    """memory_masks / key_masks (True = masked; -inf logits in the reference, rpe_transformer.py:114-119, vanilla_transformer.py:66-67)
    on the invariant RPE layer (incl. the returned score tensor) and on plain cross attention, against the oracle."""
    from oracle import se3et_oracle as O
    from se3et_amd.modules.transformer import MultiHeadAttention, RPEMultiHeadAttention
    g = torch.Generator().manual_seed(31)
    N, M, C, H = 45, 45, 64, 4
    st = _attn_state(C, False, seed=3)
    x = torch.randn(N, C, generator=g)
    emb = torch.randn(N, M, C, generator=g) * 0.5
    masks = torch.rand(M, generator=g) < 0.3
    att = RPEMultiHeadAttention(C, H, return_scores=True)
    att.load_state_dict({k[2:]: v for k, v in st.items()})
    att = att.cuda()
    with torch.no_grad():
        hidden, scores = att(x.cuda()[None], x.cuda()[None], x.cuda()[None], emb.cuda()[None], key_masks=masks.cuda()[None])
    want_h, want_s = O.rpe_attention(st, 'l.', x, x, emb, None, H, masks=masks[None, None, :])
    assert_close(hidden[0].cpu(), want_h, 1e-4, 'masked rpe attention')
    assert_close(scores[0].cpu(), want_s, 1e-4, 'masked rpe attention scores')
    assert float(scores[0][..., masks.cuda()].abs().max()) == 0.0
    cross = MultiHeadAttention(C, H)
    cross.load_state_dict({k[2:]: v for k, v in st.items() if 'proj_p' not in k})
    cross = cross.cuda()
    mem = torch.randn(M, C, generator=g)
    with torch.no_grad():
        got, _ = cross(x.cuda()[None], mem.cuda()[None], mem.cuda()[None], key_masks=masks.cuda()[None])
    q, k, v = [torch.nn.functional.linear(t, st['l.proj_%s.weight' % n], st['l.proj_%s.bias' % n]).view(-1, H, C // H).transpose(0, 1)
               for t, n in ((x, 'q'), (mem, 'k'), (mem, 'v'))]
    s = (q @ k.transpose(1, 2) / (C // H) ** 0.5).masked_fill(masks[None, None, :], float('-inf'))
    want = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(N, C)
    assert_close(got[0].cpu(), want, 1e-4, 'masked cross attention')


@pytest.mark.parametrize('N,M,C', [(382, 304, 256), (59, 53, 32), (1, 700, 128)])
def test_superpoint_scores_match_oracle(N, M, C):
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(10)
    r = torch.nn.functional.normalize(torch.randn(N, C, generator=g), dim=1)
    s = torch.nn.functional.normalize(torch.randn(M, C, generator=g) + 0.5 * r[torch.randint(0, N, (M,), generator=g)], dim=1)
    want = torch.exp(-O.pairwise_distance(r, s, normalized=True))
    want = (want / want.sum(1, keepdim=True)) * (want / want.sum(0, keepdim=True))
    got = SF.superpoint_scores(r.cuda(), s.cuda(), True).cpu()
    assert_close(got, want, 1e-4, 'superpoint scores')
    assert float((got / want - 1).abs().max()) < 1e-3


def test_weighted_procrustes_matches_oracle():
    from oracle import se3et_oracle as O
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(11)
    segs = [0, 5, 5, 40, 43, 300, 1000]                        # includes an empty and a 3-point segment
    total = segs[-1]
    src = torch.randn(total, 3, generator=g)
    ang = torch.tensor(0.7)
    R = torch.tensor([[torch.cos(ang), -torch.sin(ang), 0.], [torch.sin(ang), torch.cos(ang), 0.], [0., 0., 1.]])
    ref = src @ R.t() + torch.tensor([0.3, -0.2, 0.1]) + 0.01 * torch.randn(total, 3, generator=g)
    w = torch.rand(total, generator=g)
    offsets = torch.tensor(segs, dtype=torch.int64)
    got = SF.weighted_procrustes(src.cuda(), ref.cuda(), w.cuda(), offsets.cuda()).cpu()
    for s in range(len(segs) - 1):
        a, b = segs[s], segs[s + 1]
        if b - a >= 3:
            want = O.weighted_procrustes(src[None, a:b], ref[None, a:b], w[None, a:b])[0]
            assert_close(got[s], want, 1e-4, 'segment %d' % s)
            assert abs(float(torch.det(got[s][:3, :3])) - 1) < 1e-4
    # gated refinement step = the reference's re-scoring followed by a solve
    T0 = got[-1]
    res = torch.linalg.norm(ref - (src @ T0[:3, :3].t() + T0[:3, 3]), dim=1)
    want = O.weighted_procrustes(src[None], ref[None], (w * (res < 0.02).float())[None])[0]
    whole = torch.tensor([0, total], dtype=torch.int64)
    got2 = SF.weighted_procrustes(src.cuda(), ref.cuda(), w.cuda(), whole.cuda(), gate_transform=T0.cuda(), gate_radius=0.02)[0]
    assert_close(got2.cpu(), want, 1e-4, 'gated solve')
    votes = SF.count_inliers(src.cuda(), ref.cuda(), got.cuda(), 0.02).cpu()
    for s in (2, 4, 5):
        Ts = got[s]
        r = torch.linalg.norm(ref - (src @ Ts[:3, :3].t() + Ts[:3, 3]), dim=1)
        assert abs(int(votes[s]) - int((r < 0.02).sum())) <= 2          # boundary residuals may flip in fp32
    # reflection case: planar, mirrored correspondences must still give a proper rotation
    p = torch.randn(50, 3, generator=g); p[:, 2] = 0
    q = p.clone(); q[:, 0] = -q[:, 0]
    T = SF.weighted_procrustes(p.cuda(), q.cuda(), torch.ones(50).cuda(), torch.tensor([0, 50]).cuda())[0].cpu()
    assert abs(float(torch.det(T[:3, :3])) - 1) < 1e-4
    want = O.weighted_procrustes(p[None], q[None], torch.ones(1, 50))[0]
    assert_close(T, want, 1e-3, 'reflection case')


@pytest.mark.parametrize('N,M,K', [(2500, 382, 64), (700, 59, 64), (300, 300, 16), (5000, 7, 64), (5000, 40, 128), (3000, 263, 128)])
def test_point_to_node_partition_matches_oracle(N, M, K):
    """csrc/partition.hip against the oracle's restatement of pointcloud_partition.py:60-107 (exact: indices and masks)."""
    from oracle import se3et_oracle as O
    from se3et_amd.modules.ops import point_to_node_partition
    g = torch.Generator().manual_seed(13)
    pts = torch.rand(N, 3, generator=g) * torch.tensor([1.5, 1.2, 1.0])
    nodes = pts[torch.randperm(N, generator=g)[:M]] + 0.01 * torch.randn(M, 3, generator=g)
    want = O.point_to_node_partition(pts, nodes, K)
    got = point_to_node_partition(pts.cuda(), nodes.cuda(), K)
    p2n, masks, knn, knn_masks = [t.cpu() for t in got]
    assert torch.equal(p2n, want[0]), 'point_to_node'
    assert torch.equal(masks, want[1]), 'node_masks'
    assert torch.equal(knn_masks, want[3]), 'node_knn_masks'
    if torch.equal(knn, want[2]):
        return
    # the oracle's distances come from a matrix product, the kernel's from an fma chain: near-equal distances may swap.
    # Require: the same points per node (or, for nodes with more than K own points, equally near ones) in non-decreasing order.
    sq = O.pairwise_distance(nodes, pts)
    for m in torch.nonzero((knn != want[2]).any(1))[:, 0].tolist():
        g, w = knn[m][knn_masks[m]], want[2][m][want[3][m]]
        dg, dw = sq[m][g], sq[m][w]
        assert torch.allclose(dg, dw, rtol=1e-5, atol=1e-7), 'node %d: selected distances differ' % m
        assert bool((dg[1:] >= dg[:-1] - 1e-6 * dg[1:].clamp(min=1e-6)).all()), 'node %d: not sorted' % m
        assert bool((p2n[g] == m).all()) and len(set(g.tolist())) == len(g), 'node %d: foreign or repeated points' % m


@pytest.mark.parametrize('K', [64, 128])
def test_point_to_node_partition_stack_equals_per_cloud_calls(K):
    """se3_point_to_node_partition_stack (all clouds in one launch per kernel, global indices) against one
    se3_point_to_node_partition call per cloud."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(21)
    plen, nlen = [2500, 1800, 700, 300, 64], [382, 304, 59, 7, 90]          # the last cloud has more nodes than points
    pts = [torch.rand(n, 3, generator=g) * torch.tensor([1.5, 1.2, 1.0]) + 3.0 * c for c, n in enumerate(plen)]
    nodes = [p[torch.randperm(len(p), generator=g)[:m]] if m <= len(p) else torch.rand(m, 3, generator=g) + 3.0 * c
             for c, (p, m) in enumerate(zip(pts, nlen))]
    P, M = torch.cat(pts).cuda(), torch.cat(nodes).cuda()
    p2n, masks, knn, km = ops.point_to_node_partition_stack(P, M, plen, nlen, K)
    p0 = m0 = 0
    for p, nd in zip(pts, nodes):
        a, b, c, d = ops.point_to_node_partition(p.cuda(), nd.cuda(), K)
        n, m = len(p), len(nd)
        assert torch.equal(p2n[p0:p0 + n], a + m0)
        assert torch.equal(masks[m0:m0 + m], b)
        want = torch.where(c == n, torch.full_like(c, P.shape[0]), c + p0)
        assert torch.equal(knn[m0:m0 + m], want)
        assert torch.equal(km[m0:m0 + m], d)
        p0 += n
        m0 += m


@pytest.mark.parametrize('masked', [False, True])
def test_superpoint_scores_stack_equals_per_pair_calls(masked):
    """se3_superpoint_scores_stack (all pairs in one launch per kernel, padded rows, absent nodes = -1) against
    se3_superpoint_scores per pair on the compacted features (the reference drops empty nodes before scoring,
    superpoint_matching.py:24-29)."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(22)
    C = 256
    Ns, Ms = [382, 59, 16], [304, 53, 14]
    rows = 1010                                      # the pairs sit at arbitrary row offsets of one feature array
    feats = torch.nn.functional.normalize(torch.randn(rows, C, generator=g), dim=1).cuda()
    ref_rows, src_rows = [0, 400, 520], [700, 460, 540]
    node_masks = torch.ones(sum(Ns) + sum(Ms), dtype=torch.bool)
    if masked:
        node_masks[torch.randperm(len(node_masks), generator=g)[:60]] = False
    ref_off, src_off, o = [], [], 0
    for n, m in zip(Ns, Ms):
        ref_off.append(o)
        src_off.append(o + n)
        o += n + m
    S = ops.superpoint_scores_stack(feats, node_masks.cuda(), ref_rows, src_rows, Ns, Ms, ref_off, src_off, True)
    assert S.shape == (3, max(n * m for n, m in zip(Ns, Ms)))
    for p, (n, m) in enumerate(zip(Ns, Ms)):
        rm, sm = node_masks[ref_off[p]:ref_off[p] + n], node_masks[src_off[p]:src_off[p] + m]
        ri, si = torch.nonzero(rm)[:, 0].cuda(), torch.nonzero(sm)[:, 0].cuda()
        want = ops.superpoint_scores(feats[ref_rows[p]:ref_rows[p] + n][ri].contiguous(), feats[src_rows[p]:src_rows[p] + m][si].contiguous(), True)
        got = S[p, :n * m].view(n, m)
        sub = got[ri][:, si]
        if masked:
            assert_close(sub.cpu(), want.cpu(), 1e-6, 'stack-mode superpoint scores, pair %d' % p)
        else:
            assert torch.equal(sub, want)
        absent = torch.ones(n, m, dtype=torch.bool)
        absent[rm[:, None] & sm[None, :]] = False
        assert bool((got.cpu()[absent] == -1).all()) and bool((S[p, n * m:] == -1).all())


def test_knn3_stack_equals_per_cloud_calls():
    from se3et_amd import ops
    from se3et_amd._lib import lib, check
    g = torch.Generator().manual_seed(31)
    lens = [382, 350, 59, 1, 5, 1500, 2]
    pts = torch.cat([torch.rand(n, 3, generator=g) + 2.0 * c for c, n in enumerate(lens)]).cuda()
    got = ops.knn3_stack(pts, lens)
    o = 0
    for n in lens:
        want = torch.empty((n, 3), dtype=torch.int64, device='cuda')
        check(lib().se3_knn3(pts[o:o + n].contiguous().data_ptr(), n, want.data_ptr(), ops._stream()), 'se3_knn3')
        assert torch.equal(got[o:o + n], want)
        o += n


@pytest.mark.parametrize('N', [382, 59, 5, 1500])
def test_knn3_matches_oracle(N):
    from oracle import se3et_oracle as O
    from se3et_amd import ops
    from se3et_amd._lib import lib, check
    g = torch.Generator().manual_seed(17)
    pts = torch.rand(N, 3, generator=g) * 2.0
    dist = O.pairwise_distance(pts, pts)
    want = dist.topk(4, dim=1, largest=False)[1][:, 1:]
    p = pts.cuda()
    knn = torch.empty((N, 3), dtype=torch.int64, device='cuda')
    check(lib().se3_knn3(p.data_ptr(), N, knn.data_ptr(), None), 'se3_knn3')
    torch.cuda.synchronize()
    assert torch.equal(knn.cpu(), want)


@pytest.mark.parametrize('lengths,mode', [(((70, 61), (45, 90)), 'a_soft'), (((382, 304),), 'r_soft'), (((33, 40), (64, 64), (17, 100)), 'a_soft')])
def test_cross_attention_eq_stack_bf16x6_matches_the_single_pair_kernels(lengths, mode):
    """se3_cross_eq_stack_x6_fwd (bf16 matrix cores, six piece products) against the f32 single-pair kernels (se3_cross_eq_stats / _apply)
    pair by pair at 2e-5, and against its own f32 stack form (SE3_CROSS_EQ=f32)."""
    from se3et_amd import functional as SF
    from se3et_amd import ops, tables
    g = torch.Generator().manual_seed(len(lengths) * 7 + lengths[0][0])
    A, C, H = 6, 256, 4
    trace = torch.from_numpy(tables.trace_indices()[0]).cuda()
    pad = lambda n: (n + 31) // 32 * 32
    qs, ks, q_starts, k_starts = [], [], [], []
    rq = rk = 0
    for n, m in lengths:
        q_starts.append(rq); k_starts.append(rk)
        qs.append(torch.randn(A, n, C, generator=g)); ks.append((torch.randn(A, m, C, generator=g), torch.randn(A, m, C, generator=g)))
        rq += pad(n); rk += pad(m)
    q = torch.zeros(A, rq, C); k = torch.zeros(A, rk, C); v = torch.zeros(A, rk, C)
    for (n, m), s0, t0, qq, (kk, vv) in zip(lengths, q_starts, k_starts, qs, ks):
        q[:, s0:s0 + n] = qq; k[:, t0:t0 + m] = kk; v[:, t0:t0 + m] = vv
    q, k, v = q.cuda(), k.cuda(), v.cuda()
    vt = v.transpose(1, 2).contiguous()                                  # (A, C, rk): key-padded transposed values
    outs = {}
    for name, flag in (('bf16x6', True), ('f32', False)):
        ops.CROSS_EQ_BF16X6 = flag
        out = torch.zeros(A, rq, C, device='cuda')
        mix, w = ops.cross_attention_eq_stack(q, k, vt, q_starts, [n for n, _ in lengths], k_starts, [m for _, m in lengths], H, mode, trace, out)
        outs[name] = (out, mix, w)
    ops.CROSS_EQ_BF16X6 = True
    assert_close(outs['bf16x6'][0], outs['f32'][0], 2e-5, 'stack: bf16x6 against f32')
    eye, zero = torch.eye(C).cuda(), torch.zeros(C).cuda()
    for p, ((n, m), s0, t0) in enumerate(zip(lengths, q_starts, k_starts)):
        vt1 = SF.project_values_transposed(v[:, t0:t0 + m].contiguous(), eye, zero)
        want, want_w, want_mix = SF.cross_attention_eq(q[:, s0:s0 + n].contiguous(), k[:, t0:t0 + m].contiguous(), vt1, H, mode, trace)
        assert_close(outs['bf16x6'][0][:, s0:s0 + n], want, 2e-5, 'pair %d hidden' % p)
        assert_close(outs['bf16x6'][1][p], want_mix, 1e-5, 'pair %d mix' % p)
    # key-anchor groups (what cross_eq_groups picks for few pairs, and every divisor of A): G partial sums side by side along the channels
    picked = ops.cross_eq_groups(A, [n for n, _ in lengths], H, C, k_starts, vt)
    assert picked == 3          # (all three cases are small: 72 workgroups or fewer without groups)
    assert ops.cross_eq_groups(A, [382] * 8, H, C, [416 * i for i in range(8)], vt) == 1          # the bench batch fills the chip as it is
    for G in (2, 3, 6):
        out = torch.full((A, rq, G * C), float('nan'), device='cuda')
        mix, w = ops.cross_attention_eq_stack(q, k, vt, q_starts, [n for n, _ in lengths], k_starts, [m for _, m in lengths], H, mode, trace, out, groups=G)
        assert torch.equal(mix, outs['bf16x6'][1]) and torch.equal(w, outs['bf16x6'][2])
        for (n, m), s0 in zip(lengths, q_starts):
            got = out[:, s0:s0 + n].view(A, n, G, C).sum(2)
            assert_close(got, outs['bf16x6'][0][:, s0:s0 + n], 2e-6, '%d key-anchor groups' % G)
    W = torch.randn(64, C, generator=g).cuda()
    assert torch.equal(ops.stacked_weight(W, 3), torch.cat([W, W, W], 1)) and ops.stacked_weight(W, 3) is ops.stacked_weight(W, 3)


@pytest.mark.parametrize('rows,K,N,bias,relu', [(4096, 256, 256, True, False), (5000, 32, 64, False, False), (2049, 1024, 256, True, True),
                                                (3000, 256, 1552, True, False), (2500, 64, 96, False, True), (7777, 512, 128, False, False),
                                                (2048, 1536, 512, True, False)])
def test_linear_f16_split_has_f32_accuracy(rows, K, N, bias, relu):
    """csrc/linear_f16.hip (f16 hi / lo pieces of both operands, three products in f32) against a float64 evaluation: not worse than twice
    the library f32 GEMM's own error, and within 2e-6 of the largest output; column counts that are not multiples of 64, a row count that
    is not a multiple of the 128-row tile, bias and ReLU epilogues."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = (torch.randn(rows, K, generator=g) * torch.rand(rows, 1, generator=g) * 3).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    got = ops.linear_f16(x, w, b, relu)
    ref = torch.nn.functional.linear(x.double(), w.double(), None if b is None else b.double())
    lib32 = torch.nn.functional.linear(x, w, b)
    if relu:
        ref, lib32 = ref.clamp_min(0), lib32.clamp_min(0)
    e_new, e_lib = float((got.double() - ref).abs().max()), float((lib32.double() - ref).abs().max())
    assert e_new <= max(2 * e_lib, 2e-6 * float(ref.abs().max())), (e_new, e_lib)
    # a changed weight (in-place update bumps the version counter) is picked up
    w.mul_(0.5)
    got2 = ops.linear_f16(x, w, b, relu)
    ref2 = torch.nn.functional.linear(x.double(), w.double(), None if b is None else b.double())
    if relu:
        ref2 = ref2.clamp_min(0)
    assert float((got2.double() - ref2).abs().max()) <= 2e-6 * float(ref2.abs().max()) + 2 * e_lib


@pytest.mark.gpu
def test_weight_caches_survive_writes_behind_the_version_counter():
    """the f16 weight-piece caches are keyed by (address, shape, version).  (a) An entry must belong to the SAME tensor object
    (identity, not liveness): another tensor that inherits a freed address and an equal version must not be served the old pieces.
    (b) `.data.copy_()` bypasses the version counter: validate_weight_caches() compares the contents on the device and drops the entry;
    SE3ET.load_state_dict / ._apply clear the caches."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(256, 256, generator=g).cuda()
    w = torch.nn.Parameter((torch.randn(128, 256, generator=g) / 16).cuda(), requires_grad=False)
    ref = lambda w_: torch.nn.functional.linear(x.double(), w_.double())
    ops.clear_weight_caches()
    y0 = ops.linear_f16(x, w)
    assert float((y0.double() - ref(w)).abs().max()) < 1e-5
    # (b) a write that torch's version counter does not see
    v = w._version
    w.data.copy_(w.data * -2.0)
    assert w._version == v
    assert ops.validate_weight_caches() >= 1
    y1 = ops.linear_f16(x, w)
    assert float((y1.double() - ref(w)).abs().max()) < 1e-5
    assert ops.validate_weight_caches() == 0
    # (a) same address, same shape, same version, another tensor object
    key = (w.data_ptr(), 128, 256, w.stride(0), w.device.index)
    assert key in ops._linear_piece_cache
    entry = ops._linear_piece_cache[key]
    other = torch.nn.Parameter(torch.empty_like(w), requires_grad=False)
    other.data = w.data                          # shares the storage (same data_ptr), its own version counter (0 like w's)
    assert other.data_ptr() == w.data_ptr() and other._version == w._version
    assert entry[0]() is w and entry[0]() is not other
    w.data.copy_(w.data * 0.5)                   # stale pieces in the cache now; `other` must not be served them
    y2 = ops.linear_f16(x, other)
    assert float((y2.double() - ref(other)).abs().max()) < 1e-5
    ops.clear_weight_caches()


@pytest.mark.parametrize('rows,K,N,bias,relu', [(5000, 256, 256, True, False), (352, 256, 512, True, True), (33, 512, 256, False, False),
                                               (4224, 256, 1552, True, False), (1, 32, 7, True, True), (700, 1536, 512, True, True),
                                               (20001, 64, 100, False, False), (130, 1024, 256, True, False), (9000, 128, 32, True, True)])
def test_linear_stream_has_f32_accuracy(rows, K, N, bias, relu):
    """csrc/dense_norm.hip plain mode (se3_linear_stream: the transformer's nn.Linear layers in inference) against a float64 evaluation: not
    worse than twice the library f32 GEMM's own error; row counts from 1 to 20 000 (every tile configuration), output widths that do not fill
    a column block, bias / ReLU epilogues; strided input rows and a strided output view."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = (torch.randn(rows, K, generator=g) * torch.rand(rows, 1, generator=g) * 3).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    got = ops.linear_stream(x, w, b, relu)
    ref = torch.nn.functional.linear(x.double(), w.double(), None if b is None else b.double())
    lib32 = torch.nn.functional.linear(x, w, b)
    if relu:
        ref, lib32 = ref.clamp_min(0), lib32.clamp_min(0)
    e_new, e_lib = float((got.double() - ref).abs().max()), float((lib32.double() - ref).abs().max())
    assert e_new <= max(2 * e_lib, 2e-6 * float(ref.abs().max())), (e_new, e_lib)
    # a column block of a wider tensor as input, a column block of a wider tensor as output
    wide = torch.zeros(rows, K + 64, device='cuda')
    wide[:, 32:32 + K] = x
    out_wide = torch.full((rows, N + 12), 7.0, device='cuda')
    ops.linear_stream(wide[:, 32:32 + K], w, b, relu, out=out_wide[:, 4:4 + N])
    assert torch.equal(out_wide[:, 4:4 + N], got)
    assert bool((out_wide[:, :4] == 7.0).all()) and bool((out_wide[:, 4 + N:] == 7.0).all())


@pytest.mark.parametrize('A,R,K,N', [(6, 704, 256, 256), (1, 352, 256, 256), (6, 5632, 256, 256), (2, 96, 128, 128), (3, 32, 64, 40)])
def test_linear_stream_transposed_is_the_value_projection(A, R, K, N):
    """se3_linear_stream_transposed: (x[a] W^T + b)^T for every anchor block in one launch = project_values_transposed of the packed rows."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(A + R)
    x = torch.randn(A, R, K, generator=g).cuda()
    w, b = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(), torch.randn(N, generator=g).cuda()
    got = ops.linear_stream_transposed(x, w, b, ld=R + 8)
    ref = (torch.einsum('ark,nk->anr', x.double(), w.double()) + b.double()[None, :, None])
    assert tuple(got.shape) == (A, N, R + 8)
    assert float((got[..., :R].double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize('B,K,C,rows', [(37, 64, 256, 5000), (5, 128, 256, 900), (300, 64, 128, 70)])
def test_patch_scores_fuse_the_gathers(B, K, C, rows):
    """csrc/matching.hip patch_scores_kernel = index_select of the zero-padded fine features + einsum('bnd,bmd->bnm') / sqrt(C)
    (experiments/se3ete.3dmatch/model.py:186-203), exact f32 products."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(B + K)
    feats = torch.randn(rows, C, generator=g).cuda()
    ri = torch.randint(0, rows + 1, (B, K), generator=g).cuda()          # rows = the padding index
    si = torch.randint(0, rows + 1, (B, K), generator=g).cuda()
    padded = torch.cat((feats, torch.zeros(1, C, device='cuda')), 0).double()
    ref = torch.einsum('bnd,bmd->bnm', padded[ri], padded[si]) / C ** 0.5
    got = ops.patch_scores(feats, ri, si, 1.0 / C ** 0.5)
    assert float((got.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_anchor_mix_stack_matches_per_pair_einsum():
    """csrc/rowops.hip anchor_mix_stack_kernel = eq2inv_soft's sum_e mix[a, e] feats[e] (conditional_transformer.py:209-249) for every pair."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(3)
    starts, lengths = [0, 320, 704], [304, 382, 350]
    x = torch.randn(6, 1056, 256, generator=g).cuda()
    mixes = torch.rand(3, 6, 6, generator=g).cuda()
    got = ops.anchor_mix_stack(x, mixes, starts, lengths)
    want = torch.zeros_like(x)
    for p, (s0, n) in enumerate(zip(starts, lengths)):
        want[:, s0:s0 + n] = torch.einsum('ae,enc->anc', mixes[p], x[:, s0:s0 + n])
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float(got[:, 304:320].abs().max()) == 0.0


def _pending_reference(x, stages, seg):
    """float64: the stages v -> lrelu(v * scale + shift, slope) applied per segment."""
    x = x.double().clone()
    for aff, slope in stages:
        for s in range(len(seg) - 1):
            v = x[seg[s]:seg[s + 1]] * aff[s, 0].double() + aff[s, 1].double()
            x[seg[s]:seg[s + 1]] = torch.where(v > 0, v, v * slope)
    return x


@pytest.mark.parametrize('rows,K,N,groups,nstage,seg', [
    (6000, 64, 128, 32, 0, None), (5001, 128, 32, 32, 1, [0, 1999, 5001]), (4100, 32, 128, 32, 2, [0, 130, 2000, 4100]),
    (7000, 64, 256, 32, 2, [0, 3500, 7000]), (3000, 256, 64, 32, 1, None), (2500, 128, 512, 32, 0, [0, 700, 2500]),
    (1300, 1024, 256, 32, 1, [0, 64, 1300]), (900, 256, 1024, 32, 0, None)])
def test_dense_norm_matches_linear_then_group_norm(rows, K, N, groups, nstage, seg):
    """csrc/dense_norm.hip: y = T(x) W^T with the pending stages T applied on load and the GroupNorm statistics of y + bias taken from the
    accumulators, then group_norm_apply -- against the float64 composition T -> linear -> GroupNorm (per segment) -> LeakyReLU, with row
    counts that are not multiples of the row tile, segments shorter than a tile, 0 / 1 / 2 pending stages, every tile shape."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = (torch.randn(rows, K, generator=g) * (1 + torch.rand(1, K, generator=g) * 2) + torch.randn(1, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    gw, gb = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    segs = seg or [0, rows]
    nseg = len(segs) - 1
    stages = [((torch.randn(nseg, 2, K, generator=g) * 0.5 + torch.tensor([1.0, 0.0])[None, :, None]).cuda(), sl) for sl in (0.1, 0.2)[:nstage]]
    pend = ops.Pending(x, [a for a, _ in stages], [sl for _, sl in stages], seg)
    assert ops.dense_norm_ok(pend, w, groups)
    out = ops.dense_norm(pend, w, b, gw, gb, groups, 1e-5, seg)
    xt = _pending_reference(x, stages, segs)
    y = xt @ w.double().t()
    assert float((out.raw.double() - y).abs().max()) <= 3e-6 * float(y.abs().max())
    yb = y + b.double()
    ref = torch.empty_like(yb)
    for s in range(nseg):
        v = yb[segs[s]:segs[s + 1]].reshape(-1, groups, N // groups)
        m, var = v.mean((0, 2), keepdim=True), v.var((0, 2), unbiased=False, keepdim=True)
        ref[segs[s]:segs[s + 1]] = ((v - m) / (var + 1e-5).sqrt()).reshape(-1, N) * gw.double() + gb.double()
    ref = torch.where(ref > 0, ref, ref * 0.1)
    out.slopes[-1] = 0.1
    got = ops.group_norm_apply(out)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    if rows % 6 == 0 and N % 16 == 0 and seg is None:
        # the same pass writing the fused KPConv's gather layout
        p3 = ops.Pending(out.raw.view(rows // 6, 6, N), out.affines, out.slopes, None)
        assert torch.equal(ops.group_norm_apply(p3, blocked=True).plain().reshape(rows, N), got)
    # the statistics alone, from a pass over the raw tensor with one pending stage, agree with the epilogue's
    aff2 = ops.group_norm_stats(ops.Pending(out.raw, [], [], seg), gw, gb, groups, 1e-5, x_bias=b)
    assert torch.allclose(aff2, out.affines[0], rtol=2e-4, atol=2e-5)
    # GroupNorm of the activated tensor (the second norm of a bottleneck block), pending on a pending
    gw2, gb2 = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    aff3 = ops.group_norm_stats(out, gw2, gb2, groups, 1e-5)
    ref3 = ops.group_norm_rows(got, gw2, gb2, groups, 1e-5, 0.1, None, None, seg)
    got3 = ops.group_norm_apply(out.then(aff3, 0.1))
    assert float((got3 - ref3).abs().max()) <= 2e-5 * max(1.0, float(ref3.abs().max()))
    # shortcut branch in its pending form + final LeakyReLU: lrelu(norm(y) + norm_s(y_s))
    ys = torch.randn(rows, N, generator=g).cuda()
    affs = ops.group_norm_stats(ops.Pending(ys, [], [], seg), gw2, gb2, groups, 1e-5)
    short = ops.group_norm_rows(ys, gw2, gb2, groups, 1e-5, None, None, None, seg)
    plain = ops.Pending(out.raw, out.affines, [1.0], seg)
    ref4 = ops.group_norm_rows(out.raw, gw, gb, groups, 1e-5, 0.1, short, b, seg)
    got4 = ops.group_norm_apply(plain, ops.Pending(ys, [affs], [1.0], seg), 0.1)
    assert float((got4 - ref4).abs().max()) <= 2e-5 * max(1.0, float(ref4.abs().max()))


@pytest.mark.parametrize('rows,K,N,K2,groups,nstage,seg', [
    (6000, 32, 128, 0, 32, 2, None), (5001, 64, 256, 0, 32, 2, [0, 1999, 5001]), (4100, 32, 128, 64, 32, 2, [0, 130, 2000, 4100]),
    (7000, 64, 256, 128, 32, 2, [0, 3500, 7000]), (3000, 128, 512, 256, 32, 1, None), (1300, 256, 1024, 512, 32, 2, [0, 64, 1300]),
    (2500, 32, 64, 0, 16, 0, [0, 700, 2500]), (900, 64, 32, 32, 16, 1, None), (70, 256, 1024, 0, 32, 2, None)])
def test_block_tail_recomputed_matches_stored_form(rows, K, N, K2, groups, nstage, seg):
    """csrc/dense_norm.hip round 4: the statistics-only GEMM (dense_stats) gives the table of the storing form, and dense_residual --
    lrelu(GroupNorm(T(x) W^T + b) + R), R a tensor (K2 = 0) or a shortcut layer GroupNorm_2(x2 W2^T + b2) evaluated in the same kernel --
    equals the float64 formula and the round-3 composition (dense_norm + group_norm_apply)."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(rows + K + N + K2)
    x = (torch.randn(rows, K, generator=g) * 2).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    gw, gb = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    segs = seg or [0, rows]
    nseg = len(segs) - 1
    stages = [((torch.randn(nseg, 2, K, generator=g) * 0.5 + torch.tensor([1.0, 0.0])[None, :, None]).cuda(), sl) for sl in (0.1, 0.2)[:nstage]]
    pend = ops.Pending(x, [a for a, _ in stages], [sl for _, sl in stages], seg)
    stored = ops.dense_norm(pend, w, b, gw, gb, groups, 1e-5, seg)
    aff = ops.dense_stats(pend, w, b, gw, gb, groups, 1e-5, seg)
    assert torch.allclose(aff, stored.affines[0], rtol=1e-5, atol=1e-6)

    def gn64(y, weight, bias):
        out = torch.empty_like(y)
        for s in range(nseg):
            v = y[segs[s]:segs[s + 1]].reshape(-1, groups, N // groups)
            m, var = v.mean((0, 2), keepdim=True), v.var((0, 2), unbiased=False, keepdim=True)
            out[segs[s]:segs[s + 1]] = ((v - m) / (var + 1e-5).sqrt()).reshape(-1, N) * weight.double() + bias.double()
        return out

    main = gn64(_pending_reference(x, stages, segs) @ w.double().t() + b.double(), gw, gb)
    if K2 == 0:
        res = torch.randn(rows, N, generator=g).cuda()
        got = ops.dense_residual(pend, w, aff, residual=res, final_slope=0.1, segments=seg)
        ref = main + res.double()
        old = ops.group_norm_apply(stored, res, 0.1)
        # no residual at all
        got0 = ops.dense_residual(pend, w, aff, final_slope=0.1, segments=seg)
        ref0 = torch.where(main > 0, main, main * 0.1)
        assert float((got0.double() - ref0).abs().max()) <= 2e-5 * max(1.0, float(ref0.abs().max()))
    else:
        x2 = torch.randn(rows, K2, generator=g).cuda()
        w2 = (torch.randn(N, K2, generator=g) / K2 ** 0.5).cuda()
        b2 = torch.randn(N, generator=g).cuda()
        gw2, gb2 = (torch.rand(N, generator=g) + 0.5).cuda() * torch.where(torch.rand(N, generator=g) < 0.5, -1.0, 1.0).cuda(), torch.randn(N, generator=g).cuda()
        assert ops.norm_weight_nonzero(gw2)
        aff2 = ops.dense_stats(x2, w2, b2, gw2, gb2, groups, 1e-5, seg)
        got = ops.dense_residual(pend, w, aff, shortcut=(x2, w2, aff2), final_slope=0.1, segments=seg)
        ref = main + gn64(x2.double() @ w2.double().t() + b2.double(), gw2, gb2)
        old = ops.group_norm_apply(stored, ops.dense_norm(ops.Pending(x2, [], [], seg), w2, b2, gw2, gb2, groups, 1e-5, seg), 0.1)
    ref = torch.where(ref > 0, ref, ref * 0.1)
    scale = max(1.0, float(ref.abs().max()))
    assert float((got.double() - ref).abs().max()) <= 2e-5 * scale
    assert float((got - old).abs().max()) <= 2e-5 * scale


def test_norm_weight_nonzero_gate():
    from se3et_amd import ops
    w = torch.nn.Parameter(torch.ones(64).cuda())
    assert ops.norm_weight_nonzero(w)
    with torch.no_grad():
        w[3] = 0.0
    assert not ops.norm_weight_nonzero(w)


@pytest.mark.parametrize('P,Ns,NN,Cin,Cout,box', [(77, 90, 38, 24, 32, 0.05), (130, 130, 40, 40, 96, 0.04), (33, 64, 36, 8, 64, 0.05),
                                                  (200, 260, 48, 72, 160, 0.06), (16, 16, 16, 16, 32, 0.03), (95, 400, 64, 64, 256, 0.08),
                                                  (100, 150, 36, 128, 128, 0.05), (50, 90, 38, 256, 256, 0.05), (700, 900, 38, 256, 256, 0.12)])
def test_fused_kpconv_edge_shapes(P, Ns, NN, Cin, Cout, box):
    """The fused matrix-core kernel where its schedule has corners: more than 32 VALID neighbours per point (a second gather round), odd numbers
    of 8-channel chunks (the last producer pair is a single chunk), 3 / 5 column tiles (K-split consumers), point counts that are not
    multiples of the 16-point tile, 64 neighbours -- against the float64 formula and the f32 path."""
    from se3et_amd import functional as SF
    from se3et_amd import ops
    g = torch.Generator().manual_seed(P + Cin)
    radius, sigma = 0.0625, 0.05
    s_pts = torch.rand(Ns, 3, generator=g) * box                      # a small box: most of the NN nearest points are inside the radius
    q_pts = s_pts[torch.randperm(Ns, generator=g)[:P]].contiguous()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(min(NN, Ns), dim=1, largest=False)[1]
    if idx.shape[1] < NN:
        idx = torch.cat((idx, torch.full((P, NN - idx.shape[1]), Ns, dtype=idx.dtype)), 1)
    idx[torch.cat((d, torch.full((P, 1), 1e9)), 1).gather(1, idx) > radius ** 2] = Ns
    assert int(((idx < Ns).sum(1)).max()) > min(32, Ns - 1) or NN <= 32 or Ns <= 32
    x = torch.randn(Ns, 6, Cin, generator=g)
    st = _conv_state(Cin, Cout, radius)
    args = (x.cuda(), q_pts.cuda(), s_pts.cuda(), idx.cuda(), st['kernel_points'].cuda(), st['weights'].cuda(),
            st['kidx_rot'][:, 0, :].cuda(), st['ridx_rot'][0].cuda(), sigma)
    saved = ops.KPCONV_MATRIX_CORE
    try:
        outs = {}
        for path in (True, 'sums', False):
            ops.KPCONV_MATRIX_CORE = path
            outs[path] = SF.kpconv_inter_so3(*args).cpu()
    finally:
        ops.KPCONV_MATRIX_CORE = saved
    xs = torch.cat((x, torch.zeros(1, 6, Cin))).double()
    sp = torch.cat((s_pts, torch.full((1, 3), 1e6))).double()
    nb = sp[idx] - q_pts.double()[:, None]
    w = (1 - (nb[:, :, None] - st['kernel_points'].double()[None, None]).norm(dim=-1) / sigma).clamp(min=0)
    Fk = torch.einsum('pnk,pnac->pkac', w, xs[idx])
    W = st['weights'].double()[st['kidx_rot'][:, 0, :][:, None, :], st['ridx_rot'][0][None, :, :]]
    ref = torch.einsum('pkac,karcd->prd', Fk, W)
    e_old = float((outs[False].double() - ref).abs().max())
    for path in (True, 'sums'):
        e_new = float((outs[path].double() - ref).abs().max())
        assert e_new <= max(2 * e_old, 2e-6 * float(ref.abs().max())), (path, e_new, e_old)
    if Cin % 16 == 0:
        # the blocked feature layout [point][Cin / 16][anchor pair][16][2] (whole cache lines per gather instruction): the same arithmetic
        xb = ops.BlockedFeatures(x.view(Ns, 3, 2, Cin // 16, 16).permute(0, 3, 1, 4, 2).contiguous().cuda(), (Ns, 6, Cin))
        assert torch.equal(xb.plain().cpu(), x)
        assert torch.equal(SF.kpconv_inter_so3(xb, *args[1:]).cpu(), outs[True])
    # few tiles: the input channels are split over workgroups; the partial sums are added in a fixed order (bit-identical runs), the arrival
    # counters are back at zero for the next call, and the split form agrees with the unsplit one to round-off
    if ops.lib().se3_kpconv_fused_split_workspace_bytes(P, Cin, Cout):
        again = SF.kpconv_inter_so3(*args).cpu()
        assert torch.equal(again, outs[True])
        ops.KPCONV_SPLIT = False
        try:
            whole = SF.kpconv_inter_so3(*args).cpu()
        finally:
            ops.KPCONV_SPLIT = True
        assert float((whole - again).abs().max()) <= 5e-6 * float(ref.abs().max())
        # another split shape through the same workspace (its partial sums must not land on this shape's counters), then this one again
        other = _conv_state(64, 64, radius)
        SF.kpconv_inter_so3(torch.randn(Ns, 6, 64).cuda(), *args[1:4], other['kernel_points'].cuda(), other['weights'].cuda(),
                            other['kidx_rot'][:, 0, :].cuda(), other['ridx_rot'][0].cuda(), sigma)
        assert torch.equal(SF.kpconv_inter_so3(*args).cpu(), again)


@pytest.mark.parametrize('P,Ns,NN,Cin,Cout,box,clouds', [(77, 90, 38, 24, 32, 0.05, 1), (130, 130, 40, 40, 96, 0.04, 2), (33, 64, 36, 8, 64, 0.05, 1),
                                                         (200, 260, 48, 72, 160, 0.06, 3), (16, 16, 16, 16, 32, 0.03, 1), (95, 400, 64, 64, 256, 0.08, 2),
                                                         (100, 150, 36, 128, 128, 0.05, 1), (700, 900, 38, 256, 256, 0.12, 4), (1500, 1500, 64, 32, 64, 0.2, 3)])
def test_union_kpconv_edge_shapes(P, Ns, NN, Cin, Cout, box, clouds):
    """The union-staged fused kernel (csrc/kpconv_union.hip) at the corners of the per-lane-gather kernel's test: tables wider than 32, odd chunk
    counts, K-split consumers, point counts that are no multiple of the 16-point group, 64 neighbours -- plus what is its own: several
    clouds (every cloud starts a new group: padding positions), groups whose distinct rows exceed the 160-row cap (dense boxes: several
    passes per group), the chunked row layout, the channel split, bit-identical repeats.  Against the float64 formula and the f32 path."""
    from se3et_amd import functional as SF
    from se3et_amd import ops
    g = torch.Generator().manual_seed(P + Cin)
    radius, sigma = 0.0625, 0.05
    s_pts = torch.rand(Ns, 3, generator=g) * box
    q_pts = s_pts[torch.randperm(Ns, generator=g)[:P]].contiguous()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(min(NN, Ns), dim=1, largest=False)[1]
    if idx.shape[1] < NN:
        idx = torch.cat((idx, torch.full((P, NN - idx.shape[1]), Ns, dtype=idx.dtype)), 1)
    idx[torch.cat((d, torch.full((P, 1), 1e9)), 1).gather(1, idx) > radius ** 2] = Ns
    x = torch.randn(Ns, 6, Cin, generator=g)
    st = _conv_state(Cin, Cout, radius)
    qc = q_pts.cuda()
    args = (x.cuda(), qc, s_pts.cuda(), idx.cuda(), st['kernel_points'].cuda(), st['weights'].cuda(),
            st['kidx_rot'][:, 0, :].cuda(), st['ridx_rot'][0].cuda(), sigma)
    # the query rows as `clouds` stacked clouds of uneven sizes (tile membership only: any partition gives the same sums up to their order)
    cuts = sorted(set([0, P] + [int(P * (i + 1) ** 2 / (clouds ** 2 + 1)) for i in range(clouds - 1)]))
    lens = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    saved = (ops.KPCONV_UNION, ops.KPCONV_UNION_ALL, ops.KPCONV_MATRIX_CORE)
    try:
        ops.KPCONV_UNION = ops.KPCONV_UNION_ALL = True
        order = ops.register_point_order(qc, lens, radius / 2.5)
        assert order is not None
        o = order.cpu()
        assert sorted(o[o >= 0].tolist()) == list(range(P))                      # every point exactly once
        # the one-launch order (a workgroup per cloud sorting in LDS) is the order of the general form (keys + a stable sort + placement)
        la = ops._i64_array(lens)
        keys = torch.empty((P,), dtype=torch.int64, device='cuda')
        ops.check(ops.lib().se3_point_order_keys(qc.data_ptr(), P, la, len(lens), radius / 2.5, keys.data_ptr(), ops._stream()), 'keys')
        sk, si = torch.sort(keys, stable=True)
        order2 = torch.empty_like(order)
        ops.check(ops.lib().se3_point_order_place(sk.data_ptr(), si.data_ptr(), P, la, len(lens), order2.data_ptr(), ops._stream()), 'place')
        assert torch.equal(order, order2)
        both = ops.register_point_orders([qc, args[2]], [lens, [Ns]], [radius / 2.5, radius / 2.5])      # two stages in one launch
        assert torch.equal(both[0], order) and both[1].numel() == (Ns + 15) // 16 * 16
        assert sorted(both[1][both[1] >= 0].tolist()) == list(range(Ns))
        groups = o.view(-1, 16)
        start, row = 0, 0
        for n in lens:                                                          # every cloud's points fill whole groups of their own
            gcount = (n + 15) // 16
            mine = groups[start:start + gcount].reshape(-1)
            assert sorted(mine[mine >= 0].tolist()) == list(range(row, row + n))
            start, row = start + gcount, row + n
        assert start == groups.shape[0]
        union = SF.kpconv_inter_so3(*args).cpu()
        assert torch.equal(SF.kpconv_inter_so3(*args).cpu(), union)             # the plan is a pure function of (order, table): identical repeats
        ops.KPCONV_UNION = False
        fused = SF.kpconv_inter_so3(*args).cpu()
        ops.KPCONV_MATRIX_CORE = False
        f32 = SF.kpconv_inter_so3(*args).cpu()
        ops.KPCONV_MATRIX_CORE = saved[2]
        ops.KPCONV_UNION = True
        xs = torch.cat((x, torch.zeros(1, 6, Cin))).double()
        sp = torch.cat((s_pts, torch.full((1, 3), 1e6))).double()
        nb = sp[idx] - q_pts.double()[:, None]
        w = (1 - (nb[:, :, None] - st['kernel_points'].double()[None, None]).norm(dim=-1) / sigma).clamp(min=0)
        Fk = torch.einsum('pnk,pnac->pkac', w, xs[idx])
        W = st['weights'].double()[st['kidx_rot'][:, 0, :][:, None, :], st['ridx_rot'][0][None, :, :]]
        ref = torch.einsum('pkac,karcd->prd', Fk, W)
        e_old = float((f32.double() - ref).abs().max())
        e_new = float((union.double() - ref).abs().max())
        assert e_new <= max(2 * e_old, 2e-6 * float(ref.abs().max())), (e_new, e_old)
        assert float((union - fused).abs().max()) <= 4e-6 * float(ref.abs().max())
        # the chunked row layout [point][Cin / 8][6 anchors][8 channels]: the same arithmetic
        xb = ops.BlockedFeatures(x.view(Ns, 6, Cin // 8, 8).permute(0, 2, 1, 3).contiguous().cuda(), (Ns, 6, Cin), 2)
        assert torch.equal(xb.plain().cpu(), x)
        assert torch.equal(SF.kpconv_inter_so3(xb, *args[1:]).cpu(), union)
        # ... and a kind-1 blocked input is converted, not misread
        if Cin % 16 == 0:
            xb1 = ops.BlockedFeatures(x.view(Ns, 3, 2, Cin // 16, 16).permute(0, 3, 1, 4, 2).contiguous().cuda(), (Ns, 6, Cin), 1)
            assert torch.equal(SF.kpconv_inter_so3(xb1, *args[1:]).cpu(), union)
        G = order.numel() // 16
        if ops.lib().se3_kpconv_union_split_workspace_bytes(G, Cin, Cout):
            ops.KPCONV_SPLIT = False
            try:
                whole = SF.kpconv_inter_so3(*args).cpu()
            finally:
                ops.KPCONV_SPLIT = True
            assert float((whole - union).abs().max()) <= 5e-6 * float(ref.abs().max())
            assert torch.equal(SF.kpconv_inter_so3(*args).cpu(), union)         # the arrival counters are back at zero
        # the passes of the plan: the dense box of the last case exceeds the cap
        hit = ops._union_plan_cache.get(ops._stream().value)
        plan = hit[2].cpu().numpy()
        a16 = lambda v: (v + 15) & ~15
        nsub = plan[a16(G * 64):a16(G * 64) + 4 * G].view(np.int32)
        assert nsub.min() >= 1 and nsub.max() <= 16
        if P == 1500:
            assert nsub.max() > 1
    finally:
        ops.KPCONV_UNION, ops.KPCONV_UNION_ALL, ops.KPCONV_MATRIX_CORE = saved


@pytest.mark.parametrize('scale', [2e-38, 1e-6, 1e-3, 1.0, 3e2, 1e4, 1e8])      # (2e-38: a largest magnitude below 2^-121 -- the scale's exponent field must not overflow)
@pytest.mark.parametrize('kind', [1, 2])
def test_fused_kpconv_input_scale_over_magnitudes(scale, kind):
    """VERDICT round 4, weak 1 (the KPConv half): the fused kernels split their input into f16 hi / lo pieces -- exact only for 2^-3 <= |x| <
    65504 when x is taken as it is.  The GroupNorm apply pass that writes the kernels' input layouts also leaves the tensor's largest
    magnitude in a device word, and the kernels scale x by the power of two that brings it to [2^6, 2^7) before the split (out again in the
    epilogue): f32-level accuracy at every magnitude, both kernels (kind 1: csrc/kpconv_mfma.hip, kind 2: csrc/kpconv_union.hip)."""
    from se3et_amd import functional as SF
    from se3et_amd import ops
    g = torch.Generator().manual_seed(11)
    P = Ns = 400
    NN, Cin, Cout, radius, sigma = 34, 32, 64, 0.0625, 0.05
    s_pts = torch.rand(Ns, 3, generator=g) * 0.12
    q_pts = s_pts.clone()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(NN, dim=1, largest=False)[1]
    idx[d.gather(1, idx) > radius ** 2] = Ns
    xhat = torch.randn(Ns, 6, Cin, generator=g)
    aff = torch.stack((scale * (1 + 0.2 * torch.rand(Cin, generator=g)), scale * 0.1 * torch.randn(Cin, generator=g)))[None]      # (1 segment, 2, C)
    st = _conv_state(Cin, Cout, radius)
    qc, sc = q_pts.cuda(), s_pts.cuda()
    saved = (ops.KPCONV_UNION, ops.KPCONV_UNION_ALL)
    try:
        ops.KPCONV_UNION = ops.KPCONV_UNION_ALL = kind == 2
        if kind == 2:
            ops.register_point_order(qc, [P], radius / 2.5)
        x = ops.group_norm_apply(ops.Pending(xhat.cuda(), [aff.cuda()], [1.0], None), None, 1.0, kind)
        assert isinstance(x, ops.BlockedFeatures) and x.kind == kind and x.amax is not None
        xv = x.plain().cpu()
        assert float(x.amax.cpu()) == float(xv.abs().max())
        got = SF.kpconv_inter_so3(x, qc if kind == 2 else qc.clone(), sc if kind == 2 else sc, idx.cuda(), st['kernel_points'].cuda(), st['weights'].cuda(),
                                  st['kidx_rot'][:, 0, :].cuda(), st['ridx_rot'][0].cuda(), sigma).cpu()
    finally:
        ops.KPCONV_UNION, ops.KPCONV_UNION_ALL = saved
    xs = torch.cat((xv, torch.zeros(1, 6, Cin))).double()
    sp = torch.cat((s_pts, torch.full((1, 3), 1e6))).double()
    nb = sp[idx] - q_pts.double()[:, None]
    w = (1 - (nb[:, :, None] - st['kernel_points'].double()[None, None]).norm(dim=-1) / sigma).clamp(min=0)
    Fk = torch.einsum('pnk,pnac->pkac', w, xs[idx])
    W = st['weights'].double()[st['kidx_rot'][:, 0, :][:, None, :], st['ridx_rot'][0][None, :, :]]
    ref = torch.einsum('pkac,karcd->prd', Fk, W)
    assert torch.isfinite(got).all()
    assert float((got.double() - ref).abs().max()) <= 4e-6 * float(ref.abs().max()), (scale, kind)


@pytest.mark.parametrize('scale', [1e-6, 1e-3, 1.0, 3e2, 1e4, 1e8])
@pytest.mark.parametrize('A', [1, 6])
def test_rpe_logits_query_scale_over_magnitudes(scale, A):
    """VERDICT round 4, weak 1 (the logits kernel's half): csrc/attention.hip rpe_bias_kernel multiplies f16 hi / lo pieces of the folded
    queries; every query point's rows are scaled by ONE power of two before the split (largest magnitude -> [2^6, 2^7), nothing inside
    [2^-4, 2^7)) and the logits are scaled back in the tile epilogue: f32-level accuracy for queries of any magnitude -- 1e8 used to be Inf,
    1e-6 used to keep 11 bits.  Rows of very different magnitude inside one launch (every point its own scale)."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(5)
    N, C, H = 75, 256, 4
    emb = torch.randn(N, N, C, generator=g)
    proj = torch.randn(A, N, H * C + 4 * H, generator=g) * scale      # (qp and qe are column blocks of ONE projection, as in the model)
    proj[:, ::7] *= 1e-3                                    # some query points three orders of magnitude below the others
    pc = proj.cuda()
    qp, qe, eq = proj[..., :H * C], None, None
    if A == 6:
        qe = proj[..., H * C:]
        eq = torch.randn(A, N, N, 4, generator=g)
    got = ops.rpe_bias(pc[..., :H * C], pc[..., H * C:] if A == 6 else None, emb.cuda(), eq.cuda() if eq is not None else None, H).cpu()[:, :, :N]
    want = torch.einsum('anhc,nmc->ahnm', qp.double().view(A, N, H, C), emb.double())
    if A == 6:
        want = want + torch.einsum('anhe,anme->ahnm', qe.double().view(A, N, H, 4), eq.double())
    want = want.reshape(A * H, N, N)
    assert torch.isfinite(got).all()
    # per query point: its logits against ITS magnitude (a point 1e-3 below the rest keeps its own 22 bits)
    err = (got.double() - want).abs().amax(dim=(0, 2))
    ref = want.abs().amax(dim=(0, 2))
    assert float((err / ref).max()) <= 3e-6, (scale, A, float((err / ref).max()))


def test_magnitude_word_of_blocked_features_expires_with_its_ring():
    """ops._amax_slot hands out words of a per-stream ring that is cleared every 4096 apply passes: a BlockedFeatures kept beyond that must not
    be scaled by a word that describes another tensor by then (ops._amax_live: None -> the features are split as they are)."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(40, 6, 16, generator=g).cuda()
    aff = torch.stack((torch.ones(16), torch.zeros(16)))[None].cuda()
    first = ops.group_norm_apply(ops.Pending(x, [aff], [1.0], None), None, 1.0, 2)
    assert ops._amax_live(first) is first.amax and float(first.amax.cpu()) == float(x.abs().max().cpu())
    ring = first.amax_tag[0]
    ring[1] = 4096                                                    # the ring is about to wrap
    second = ops.group_norm_apply(ops.Pending(x * 3, [aff], [1.0], None), None, 1.0, 2)
    assert ops._amax_live(second) is second.amax and float(second.amax.cpu()) == float((x * 3).abs().max().cpu())
    # the wrap zeroed the WHOLE ring: the first object's word reads 0 from now on (no range protection) -- it is dead at once, not only
    # when the cursor reaches its position
    assert second.amax_tag[2] == 0 and first.amax_tag[2] > 0 and ops._amax_live(first) is None
    ring[1], ring[2] = 5, ring[2] + 1                                 # a second wrap
    assert ops._amax_live(second) is None


def _magnitude_stack(g, A, C, lengths, alphas):
    """Packed q / k / v of a stack of clouds (32-row padded starts): cloud c's queries times alphas[c] and 2^(-3..3) per row, its keys
    divided by alphas[c] and times 2^(-3..3) per row (rows of very different size inside one 8-row block), its value channels times
    10^(-8..8) each -- the logits stay of order one, every operand leaves the f16 range.  Returns q, k, v (A, R, C), starts, beta (n, C)."""
    pad = lambda n: (n + 31) // 32 * 32
    starts, r = [], 0
    for n in lengths:
        starts.append(r); r += pad(n)
    q = torch.zeros(A, r, C); k = torch.zeros(A, r, C); v = torch.zeros(A, r, C)
    beta = 10.0 ** (torch.rand(len(lengths), C, generator=g) * 16 - 8)
    for c, (n, s0) in enumerate(zip(lengths, starts)):
        row_q = 2.0 ** torch.randint(-3, 4, (n, 1), generator=g).float()
        row_k = 2.0 ** torch.randint(-3, 4, (n, 1), generator=g).float()
        q[:, s0:s0 + n] = torch.randn(A, n, C, generator=g) * 0.3 * row_q * alphas[c]
        k[:, s0:s0 + n] = torch.randn(A, n, C, generator=g) * 0.3 * row_k / alphas[c]
        v[:, s0:s0 + n] = torch.randn(A, n, C, generator=g) * beta[c]
    return q, k, v, starts, beta


def test_attention_stack_operand_scales_over_magnitudes():
    """The f16 attention kernel's operands carry their own powers of two (csrc/attention.hip: x6_split_kernel -- a query row per head, 8 key
    rows per head, a value channel per cloud): queries at 1e-12 / 3e7 / 1 by cloud against keys at the inverse, rows of one key block 64x
    apart, value channels 1e-8 ... 1e8 -- against a float64 evaluation at the f32 tolerance, channel by channel; nothing is clamped.  A
    non-finite key is counted (ops.attention_saturated) and PROPAGATES as the reference's matmul + softmax would: every
    query of that (cloud, anchor, head) is NaN, the other anchors, heads and clouds bit for bit."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(21)
    A, H, C, lengths, alphas = 6, 4, 256, (70, 64, 33), (1e-12, 3e7, 1.0)
    q, k, v, starts, beta = _magnitude_stack(g, A, C, lengths, alphas)
    biases, offsets, off = [], [], 0
    for n in lengths:
        mp = (n + 31) // 32 * 32
        b = torch.randn(A * H, n, mp, generator=g)
        biases.append(b); offsets.append(off); off += b.numel()
    bias = torch.cat([b.reshape(-1) for b in biases]).cuda()
    # the rows / value columns between a cloud's end and the next cloud's start belong to nobody: whatever they hold must not matter
    for n, s0, s1 in zip(lengths, starts, starts[1:] + [q.shape[1]]):
        k[:, s0 + n:s1] = float('nan')
        v[:, s0 + n:s1] = float('nan')
    vt = v.transpose(1, 2).contiguous().cuda()

    def run(kk):
        out = torch.zeros(A, q.shape[1], C, device='cuda')
        ops.attention_stack(q.cuda(), kk.cuda(), vt, bias, offsets, starts, list(lengths), starts, list(lengths), H, out)
        return out.cpu()
    ops.attention_saturated(reset=True)
    got = run(k)
    assert ops.attention_saturated() == 0
    D = C // H
    for c, (n, s0) in enumerate(zip(lengths, starts)):
        qd = q[:, s0:s0 + n].double().view(A, n, H, D); kd = k[:, s0:s0 + n].double().view(A, n, H, D)
        vd = (v[:, s0:s0 + n] / beta[c]).double().view(A, n, H, D)
        sc = (torch.einsum('anhd,amhd->ahnm', qd, kd) + biases[c].double().view(A, H, n, -1)[..., :n]) / D ** 0.5
        want = torch.einsum('ahnm,amhd->anhd', sc.softmax(-1), vd).reshape(A, n, C)
        assert_close((got[:, s0:s0 + n] / beta[c]).double(), want, 1e-4, 'cloud %d (queries x %g)' % (c, alphas[c]))
    bad = k.clone()
    bad[2, starts[1] + 5, 7] = float('inf')
    out = run(bad)
    assert ops.attention_saturated() > 0
    other = [a for a in range(A) if a != 2]
    assert torch.equal(out[other], got[other]) and torch.equal(out[:, :starts[1]], got[:, :starts[1]])
    assert torch.equal(out[:, starts[2]:starts[2] + lengths[2]], got[:, starts[2]:starts[2] + lengths[2]])
    hit = out[2, starts[1]:starts[1] + lengths[1]].view(lengths[1], H, D)              # channel 7 = head 0
    assert bool(torch.isnan(hit[:, 0]).all()), 'an Inf key must reach every query of its head as NaN'
    assert torch.equal(hit[:, 1:], got[2, starts[1]:starts[1] + lengths[1]].view(lengths[1], H, D)[:, 1:])


def test_cross_attention_eq_stack_operand_scales_over_magnitudes():
    """The same for the equivariant cross attention's f16 form (queries split ahead of the kernel with a scale per row and head): against its
    own f32 stack form on the same operands, channel by channel."""
    from se3et_amd import ops, tables
    g = torch.Generator().manual_seed(22)
    A, H, C, lengths, alphas = 6, 4, 256, (70, 45, 33), (1e-9, 2e6, 1.0)
    trace = torch.from_numpy(tables.trace_indices()[0]).cuda()
    q, k, v, starts, beta = _magnitude_stack(g, A, C, lengths, alphas)
    q, k, vt = q.cuda(), k.cuda(), v.transpose(1, 2).contiguous().cuda()
    outs = {}
    ops.attention_saturated(reset=True)
    for name, flag in (('f16', True), ('f32', False)):
        ops.CROSS_EQ_BF16X6 = flag
        out = torch.zeros(A, q.shape[1], C, device='cuda')
        mix, w = ops.cross_attention_eq_stack(q, k, vt, starts, list(lengths), starts, list(lengths), H, 'a_soft', trace, out)
        outs[name] = (out.cpu(), mix.cpu())
    ops.CROSS_EQ_BF16X6 = True
    assert ops.attention_saturated() == 0
    assert_close(outs['f16'][1], outs['f32'][1], 1e-5, 'mixing weights')
    for c, (n, s0) in enumerate(zip(lengths, starts)):
        assert_close(outs['f16'][0][:, s0:s0 + n] / beta[c], outs['f32'][0][:, s0:s0 + n] / beta[c], 2e-5, 'pair %d (queries x %g)' % (c, alphas[c]))


def test_neighbor_table_trim_marks_each_pairs_surplus_columns():
    """csrc/radius_neighbors.hip: the stacked neighbour table cut to the batch's width with the columns past every PAIR's own width set to
    -1 -- against the column copy + per-pair strided fill it replaces."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(3)
    rows_per_pair, W, width = [700, 13, 1200, 1], 38, 31
    widths = [31, 20, 36, 7]
    full = torch.randint(0, 5000, (sum(rows_per_pair), W), generator=g).cuda()
    ends = list(np.cumsum(rows_per_pair))
    got = ops.neighbor_table_trim(full, width, ends, widths)
    want = full[:, :width].clone()
    row = 0
    for n, w in zip(rows_per_pair, widths):
        if w < width:
            want[row:row + n, w:] = -1
        row += n
    assert torch.equal(got, want)


def test_kpconv_neighbor_table_is_shared_only_for_the_same_geometry():
    """ops._kpconv_neighbor_table: the layers of a stage reuse the table; other points, an in-place change or another extent rebuild it."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(4)
    s = torch.rand(300, 3, generator=g).cuda() * 0.2
    q = s[:120].contiguous()
    idx = ((q[:, None] - s[None]) ** 2).sum(-1).topk(24, largest=False)[1].contiguous()
    kp = _conv_state(16, 32, 0.0625)['kernel_points'].cuda()
    st = ops._stream()
    t0 = ops._kpconv_neighbor_table(q, s, idx, kp, 0.05, 120, 300, 24, st)
    assert ops._kpconv_neighbor_table(q, s, idx, kp, 0.05, 120, 300, 24, st) is t0
    assert ops._kpconv_neighbor_table(q, s, idx, kp, 0.06, 120, 300, 24, st) is not t0
    t1 = ops._kpconv_neighbor_table(q, s, idx, kp, 0.05, 120, 300, 24, st)
    q.mul_(1.0001)                                                   # same tensor, new values
    t2 = ops._kpconv_neighbor_table(q, s, idx, kp, 0.05, 120, 300, 24, st)
    assert t2 is not t1 and not torch.equal(t2, t1)
    q2 = q.clone()
    t3 = ops._kpconv_neighbor_table(q2, s, idx, kp, 0.05, 120, 300, 24, st)
    assert t3 is not t2
    # the kernel points enter by value: another layer's copy of the same 15 points shares the table, other points do not
    assert ops._kpconv_neighbor_table(q2, s, idx, kp.clone(), 0.05, 120, 300, 24, st) is t3
    assert ops._kpconv_neighbor_table(q2, s, idx, kp * 1.01, 0.05, 120, 300, 24, st) is not t3


@pytest.mark.parametrize('C', [256, 128])
@pytest.mark.parametrize('lengths', [[382, 350, 13, 129], [382], [75, 402], [33, 1]])
def test_gram_stack_kernel_matches_the_library_path(C, lengths):
    """csrc/attention.hip gram_stack_kernel (X_p^T X_p per anchor and pair straight from the packed rows) against index_select + mask +
    batched GEMM, ragged lengths incl. one that is not a multiple of the 8-row step; one or two pairs run the 32 x 32-block kernel (four times
    as many waves with a quarter of the MFMA chain each), four pairs the 64 x 64 one."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(C)
    starts, r = [], 0
    for n in lengths:
        starts.append(r)
        r += (n + 31) // 32 * 32
    x = torch.randn(6, r, C, generator=g).cuda()
    got = ops._gram_per_pair(x, starts, lengths)
    ops.GRAM_KERNEL = False
    try:
        want = ops._gram_per_pair(x, starts, lengths)
    finally:
        ops.GRAM_KERNEL = True
    ref = torch.stack([torch.stack([x[a, s:s + n].double().t() @ x[a, s:s + n].double() for s, n in zip(starts, lengths)]) for a in range(6)])
    e_new, e_lib = float((got.double() - ref).abs().max()), float((want.double() - ref).abs().max())
    assert e_new <= max(2 * e_lib, 2e-6 * float(ref.abs().max())), (e_new, e_lib)
    assert torch.equal(ops._gram_per_pair(x, starts, lengths), got)


@pytest.mark.gpu
@pytest.mark.parametrize('rows', [6, 42, 48, 54, 378, 384, 390, 6000, 6144 + 6, 100002])
@pytest.mark.parametrize('channels', [32, 256])
def test_group_norm_statistics_pass_handles_every_remainder(rows, channels):
    """The statistics pass keeps two batches of eight rows per lane in flight and clamps the last one (csrc/rowops.hip: gn_partial4_kernel):
    row counts around every batch boundary, with and without a pending stage, against torch's group_norm per segment."""
    from se3et_amd import ops
    torch.manual_seed(rows + channels)
    dev = torch.device('cuda')
    x = torch.randn(rows, channels, device=dev) * 2 + 0.5
    w = torch.rand(channels, device=dev) + 0.5
    b = torch.randn(channels, device=dev)
    cut = rows // 2 // 6 * 6
    segs = [None] if cut == 0 else [None, [0, cut, rows]]
    for seg in segs:
        bounds = [0, rows] if seg is None else seg
        aff = ops.group_norm_stats(x, w, b, 32, 1e-5, segments=seg)
        for i in range(len(bounds) - 1):
            xs = x[bounds[i]:bounds[i + 1]]
            want = torch.nn.functional.group_norm(xs.t()[None], 32, w, b, 1e-5)[0].t()
            got = xs * aff[i, 0] + aff[i, 1]
            assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), (rows, channels, seg, i)
        # through one pending stage: statistics of lrelu(norm_1(x)) without materialising it
        pend = ops.Pending(x, [aff], [0.1], seg)
        aff2 = ops.group_norm_stats(pend, w, b, 32, 1e-5)
        for i in range(len(bounds) - 1):
            xs = torch.nn.functional.leaky_relu(x[bounds[i]:bounds[i + 1]] * aff[i, 0] + aff[i, 1], 0.1)
            want = torch.nn.functional.group_norm(xs.t()[None], 32, w, b, 1e-5)[0].t()
            got = xs * aff2[i, 0] + aff2[i, 1]
            assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), (rows, channels, seg, i, 'pending')


@pytest.mark.gpu
@pytest.mark.parametrize('scale', [1e-30, 1e-6, 1e-4, 1e-2, 1.0, 1e2, 1.2e4, 1e8, 1e30])
def test_f16_split_accuracy_over_input_magnitudes(scale):
    """VERDICT round 4 (weak 1): the f16 hi + lo split of the ACTIVATIONS used to be exact only for 2^-3 <= |x| < 65504 (f16 accuracy at 1e-4,
    Inf above 65504) and nothing guarded that window.  The dense kernel now scales every ROW by a power of two taken from its first 32
    values before the split and takes the scale out of the accumulators (csrc/dense_norm.hip: row_exp): f32 accuracy (<= 2e-6 of the
    largest output, the bound the unscaled split met at |x| ~ 1) at EVERY magnitude a float32 product itself survives."""
    from se3et_amd import ops
    torch.manual_seed(3)
    dev = torch.device('cuda')
    x = torch.randn(4096, 256, device=dev) * scale
    w = torch.randn(256, 256, device=dev) / 16
    ref = x.double() @ w.double().t()
    ops.dense_saturated_rows()
    for out in (ops.linear_stream(x, w), ops.linear_stream_transposed(x.view(4, 1024, 256), w, None).transpose(1, 2).reshape(4096, 256)):
        assert bool(torch.isfinite(out).all())
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-6, (scale, err)
    assert ops.dense_saturated_rows() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('rows,K,N', [(4096, 256, 256), (1000, 64, 128), (777, 512, 64), (300, 1536, 32)])
def test_f16_split_row_scales_with_mixed_magnitudes(rows, K, N):
    """The row scales of the dense kernel where they matter: (a) every row its own magnitude, 1e-6 ... 1e8 side by side -- each ROW of the
    product within 2e-6 of ITS largest entry (rows do not influence each other); (b) magnitudes that grow along the input channels inside
    the row's headroom (2^8 above its first 32 values): exact; beyond it: clamped, finite and COUNTED (se3_debug_dense_saturated_rows);
    rows holding a NaN / Inf come out as NaN (the reference's matmul propagates them) and leave their neighbours untouched;
    (c) all-zero rows, rows of denormals, rows that start with 32 zeros."""
    from se3et_amd import ops
    torch.manual_seed(rows + K)
    dev = torch.device('cuda')
    w = torch.randn(N, K, device=dev) / 16
    rel = lambda out, ref: (out.double() - ref).abs().amax(1) / ref.abs().amax(1).clamp_min(1e-300)
    # (a) one magnitude per row
    ops.dense_saturated_rows()
    mag = 10.0 ** torch.randint(-6, 9, (rows,), device=dev).float()
    x = torch.randn(rows, K, device=dev) * mag[:, None]
    out, ref = ops.linear_stream(x, w), x.double() @ w.double().t()
    assert float(rel(out, ref).max()) <= 2e-6, ('per-row magnitudes', float(rel(out, ref).max()))
    assert ops.dense_saturated_rows() == 0
    # (b) magnitude growing along K: 2^6 above the first 32 values (inside the headroom of 2^8)
    x = torch.randn(rows, K, device=dev)
    x[:, 32:] *= 2.0 ** 6
    out, ref = ops.linear_stream(x, w), x.double() @ w.double().t()
    assert float(rel(out, ref).max()) <= 2e-6, ('growing magnitudes', float(rel(out, ref).max()))
    assert ops.dense_saturated_rows() == 0
    # ... and far beyond it in a few rows, NaN / Inf in others: those rows are clamped and counted, every OTHER row is exact
    bad = torch.zeros(rows, dtype=torch.bool, device=dev)
    bad[7::50] = True
    x = torch.randn(rows, K, device=dev)
    xb = x.clone()
    xb[bad, 32:] *= 2.0 ** 30
    xb[3, K - 1] = float('nan')
    xb[11, 40 % K] = float('inf')
    xb[12, 0] = float('-inf')
    xb[13, 5] = float('nan')
    nonfinite = torch.zeros_like(bad)
    nonfinite[3] = nonfinite[11] = nonfinite[12] = nonfinite[13] = True
    out, ref = ops.linear_stream(xb, w), x.double() @ w.double().t()
    assert float(rel(out[~(bad | nonfinite)], ref[~(bad | nonfinite)]).max()) <= 2e-6
    assert bool(torch.isfinite(out[bad & ~nonfinite]).all())          # (finite overflow of the headroom: clamped, finite -- and counted)
    assert bool(torch.isnan(out[nonfinite]).all())                    # (NaN / Inf in: NaN out, the whole row)
    assert ops.dense_saturated_rows() >= int((bad & ~nonfinite).sum())   # (events: a row counts once per K-step and column block it saturates in)
    # (c) zeros, denormals, rows that start with 32 zeros (no scale can be taken from them: unscaled)
    x = torch.randn(rows, K, device=dev)
    x[5:29] = 0
    x[40:48] = torch.randn(8, K, device=dev) * 1e-41
    if K > 32:
        x[60:70, :32] = 0
    out, ref = ops.linear_stream(x, w), x.double() @ w.double().t()
    assert float(out[5:29].abs().max()) == 0.0
    assert float((out.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert float((out[40:48].double() - ref[40:48]).abs().max()) <= 1.2e-38          # (rows of denormals: treated as zeros, off by less than the smallest normal number)
    assert float(rel(out[60:70], ref[60:70]).max()) <= 2e-6
    assert ops.dense_saturated_rows() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('rows,K,N', [(4122, 128, 512), (4122, 128, 256), (2000, 64, 128), (1500, 256, 64), (900, 512, 32), (4122, 128, 1024)])
def test_a_scaled_row_leaves_the_other_rows_of_its_tile_alone(rows, K, N):
    """Round 6 (found on synthetic pair c2_5k #1023, encoder4_1.unary2: 4122 x 128 -> 512): ONE row of a tile whose first 32 values are all
    below 2^-4 takes its own power-of-two scale; the epilogue that takes the scale out again must leave every OTHER row of the tile as it is.
    (It scaled accumulator row 1 of the tile by a stale word: Inf in one row of the layer's output, clamped unnoticed by the next layer's
    f16 split until round 6 let non-finite values propagate.)  Ordinary rows with a few tiny-start rows at assorted tile positions, the plain
    layer and the layer fused with its GroupNorm statistics (pending input norms: the backbone's form): every row against float64."""
    from se3et_amd import ops
    torch.manual_seed(rows + N)
    dev = torch.device('cuda')
    x = torch.randn(rows, K, device=dev)
    tiny = [19, 20, 36, 51, 83, 100, rows // 2 + 3, rows - 7, rows - 1]
    for r in tiny:
        x[r, :32] = torch.randn(32, device=dev) * 0.01                  # first K-step below 2^-4, the rest of the row ordinary
    w = torch.randn(N, K, device=dev) / K ** 0.5
    want = x.double() @ w.double().t()
    ops.dense_saturated_rows(reset=True)
    out = ops.linear_stream(x, w)
    assert bool(torch.isfinite(out).all())
    err = (out.double() - want).abs().amax(1) / want.abs().amax()
    assert float(err.max()) <= 2e-6, ('plain layer', int(err.argmax()), float(err.max()))
    if ops.dense_norm_ok(x, w, max(1, N // 32)):
        # the fused form: two pending input stages (scale / shift + LeakyReLU), statistics of the output's GroupNorm from the accumulators
        a1 = torch.stack((1 + 0.1 * torch.rand(K, device=dev), 0.05 * torch.randn(K, device=dev)))[None]
        a2 = torch.stack((1 + 0.1 * torch.rand(K, device=dev), 0.05 * torch.randn(K, device=dev)))[None]
        t = x * a1[0, 0] + a1[0, 1]
        t = torch.where(t > 0, t, t * 0.1)
        t = t * a2[0, 0] + a2[0, 1]
        t = torch.where(t > 0, t, t * 0.1)
        for r in tiny:                                                   # the rows the KERNEL sees as tiny: after the pending stages
            x[r, :32] = ((torch.randn(32, device=dev) * 0.01 - a2[0, 1, :32]) / a2[0, 0, :32] - a1[0, 1, :32]) / a1[0, 0, :32]
        t = x * a1[0, 0] + a1[0, 1]
        t = torch.where(t > 0, t, t * 0.1)
        t = t * a2[0, 0] + a2[0, 1]
        t = torch.where(t > 0, t, t * 0.1)
        want = t.double() @ w.double().t()
        P = ops.Pending(x, [a1, a2], [0.1, 0.1], None)
        gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        res = ops.dense_norm(P, w, None, gamma, beta, max(1, N // 32) if N >= 32 else 1, 1e-5)
        raw = (res.raw if isinstance(res, ops.Pending) else res[0]).reshape(rows, N)
        assert bool(torch.isfinite(raw).all())
        err = (raw.double() - want).abs().amax(1) / want.abs().amax()
        assert float(err.max()) <= 2e-6, ('fused with statistics', int(err.argmax()), float(err.max()))
    assert ops.dense_saturated_rows() == 0


@pytest.mark.gpu
def test_dense_inputs_of_the_c2_forward_stay_in_the_split_range():
    """Every tensor that the inference forward of the headline configuration hands to an f16-split kernel (dense layers plain / with pending
    norms, KPConv) lies inside the range in which the split is f32-accurate: far from the f16 overflow (largest magnitude below 65504 / 64 --
    the KPConv sums ~40 weighted neighbours) and not a tensor of tiny numbers (RMS above 2^-7: the bulk of the values have normal lo pieces
    or contribute below the f32 round-off of the larger ones)."""
    from se3et_amd import cdriver, ops
    from se3et_amd.batched import forward_pairs
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    cfg = make_cfg('se3ete')
    model = load_synthetic_weights(create_model(cfg)).cuda().eval()
    b = cfg.backbone
    ref, src, _ = make_pair('c2_5k', index=0)
    pts = torch.from_numpy(np.concatenate([ref, src], 0)).cuda()
    lens = torch.tensor([len(ref), len(src)])
    seen = []

    def concrete(x):
        if isinstance(x, ops.Pending):
            v = x.raw
            segs = x.segments
            for aff, slope in zip(x.affines, x.slopes):
                if aff.shape[0] == 1:
                    v = v * aff[0, 0] + aff[0, 1]
                else:
                    v = torch.cat([v[segs[i]:segs[i + 1]] * aff[i, 0] + aff[i, 1] for i in range(aff.shape[0])], 0) if v.dim() == 2 else v
                v = torch.nn.functional.leaky_relu(v, slope) if slope != 1.0 else v
            return v
        if isinstance(x, ops.BlockedFeatures):
            return x.plain()
        return x

    def wrap(name):
        orig = getattr(ops, name)

        def f(x, *a, **k):
            v = concrete(x).detach().float()
            seen.append((name, tuple(v.shape), float(v.abs().max()), float(v.pow(2).mean().sqrt())))
            return orig(x, *a, **k)
        setattr(ops, name, f)
        return orig
    names = ('linear_stream', 'linear_stream_transposed', 'dense_norm', 'dense_stats', 'dense_residual', 'kpconv_inter_so3')
    saved = {n: wrap(n) for n in names}
    enabled, cdriver.ENABLED = cdriver.ENABLED, False          # (the C-issued transformer would bypass the Python front ends hooked here)
    try:
        with torch.no_grad():
            data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
            data['features'] = torch.ones((pts.shape[0], 1), device='cuda')
            forward_pairs(model, data)
    finally:
        cdriver.ENABLED = enabled
        for n, f in saved.items():
            setattr(ops, n, f)
    assert len(seen) > 100
    worst_max = max(s[2] for s in seen)
    worst_rms = min(s[3] for s in seen)
    assert worst_max < 65504.0 / 64, sorted(seen, key=lambda s: -s[2])[:3]
    assert worst_rms > 2.0 ** -7, sorted(seen, key=lambda s: s[3])[:3]


@pytest.mark.gpu
@pytest.mark.parametrize('nn,width', [(1, 6), (7, 24), (8, 96), (9, 100), (16, 192), (22, 192), (38, 768), (64, 3072)])
def test_neighbor_max_pool_with_padded_and_absent_entries(nn, width):
    """The pool compacts the real neighbours before it gathers (csrc/rowops.hip): rows whose entries are all padding (index n: the zero row ->
    0), all absent (-1 beyond the pair's own table width -> -inf never wins against a padded 0, and a row of only absent columns does not
    occur: column 0 is the point itself), mixtures, real counts on both sides of a multiple of 8, float4 and scalar widths -- against a
    gather + amax."""
    from se3et_amd import ops
    g = torch.Generator().manual_seed(nn * 1000 + width)
    n, m = 300, 257
    x = torch.randn(n, width, generator=g)
    idx = torch.randint(0, n, (m, nn), generator=g)
    kind = torch.rand(m, nn, generator=g)
    idx[kind < 0.25] = n                                    # padded entries
    idx[kind > 0.85] = -1                                   # absent columns
    idx[:, 0] = torch.randint(0, n, (m,), generator=g)      # column 0 is always a real neighbour ...
    idx[0] = n                                              # ... except in the rows built here: all padding,
    idx[1, 1:] = -1                                         # one real entry and absent columns,
    if nn >= 8:
        idx[2, :8] = torch.randint(0, n, (8,), generator=g); idx[2, 8:] = n     # exactly eight real entries
    xp = torch.cat((x, torch.zeros(1, width)), 0)
    want = xp[idx.clamp(min=0)].masked_fill((idx < 0)[:, :, None], float('-inf')).amax(1)
    got = ops.neighbor_max_pool(x.cuda(), idx.cuda()).cpu()
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('rows', [1, 63, 64, 65, 700, 10176, 10240, 10304])
@pytest.mark.parametrize('K,N', [(32, 64), (256, 192), (256, 256), (512, 320), (64, 1552)])
def test_linear_stream_tile_policies_agree_with_float64(rows, K, N):
    """Row counts on both sides of the tile-policy thresholds of the plain dense layer (64 x 64 / 64 x 128 / 64 x 256 tiles by workgroup
    count, csrc/dense_norm.hip: linear_stream), ragged last tiles, output widths that are not multiples of the tile: against float64."""
    from se3et_amd import ops
    torch.manual_seed(rows + K + N)
    dev = torch.device('cuda')
    x = torch.randn(rows, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    for relu in (False, True):
        out = ops.linear_stream(x, w, b, relu=relu)
        ref = x.double() @ w.double().t() + b.double()
        ref = ref.clamp_min(0) if relu else ref
        assert float((out.double() - ref).abs().max()) <= 3e-6 * max(1.0, float(ref.abs().max())), (rows, K, N, relu)
