"""GPU: the training step (BASELINE.json configs[4] pieces) -- autograd through the HIP ops against the genuine reference's
forward + OverallLoss + backward + Adam step on the micro models (tests/golden/train_micro_*.npz).  Tolerances: losses 1e-4
relative, gradient norms 1e-3 relative."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _setup(variant, fixture, golden_dir):
    from se3et_amd.data import registration_collate_fn_stack_mode
    from se3et_amd.model import create_model, make_cfg
    g = np.load(golden_dir + '/' + fixture)
    w = np.load(golden_dir + '/' + fixture.replace('train_', ''))                # reference-initialised weights (seed 0)
    cfg = make_cfg(variant)
    model = create_model(cfg)
    model.load_state_dict({k[3:]: torch.from_numpy(w[k]) for k in w.files if k.startswith('sd/')}, strict=True)
    model = model.cuda().train()
    d = dict(ref_points=g['ref'], src_points=g['src'], ref_feats=np.ones((len(g['ref']), 1), np.float32),
             src_feats=np.ones((len(g['src']), 1), np.float32), transform=g['transform'])
    b = cfg.backbone
    dd = registration_collate_fn_stack_mode([d], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    return g, cfg, model, dd


def _check_training_step(g, cfg, model, dd, grad_tol=1e-3, full_tol=2e-3):
    from se3et_amd.training import OverallLoss, make_optimizer
    targets = (torch.from_numpy(g['target/ref']).long(), torch.from_numpy(g['target/src']).long())
    out = model(dd, train=True, targets=targets)
    # ground-truth superpoint correspondences: same pairs, same overlaps
    want = {(int(a), int(b)): float(o) for (a, b), o in zip(g['gt_node_corr_indices'], g['gt_node_corr_overlaps'])}
    got = {(int(a), int(b)): float(o) for (a, b), o in zip(out['gt_node_corr_indices'].tolist(), out['gt_node_corr_overlaps'].tolist())}
    assert set(got) == set(want)
    assert max(abs(got[k] - want[k]) for k in want) < 1e-6
    assert tuple(out['matching_scores'].shape) == tuple(g['matching_scores_shape'])
    losses = OverallLoss(cfg)(out, dd)
    for k in ('loss', 'c_loss', 'f_loss'):
        assert abs(float(losses[k]) - float(g['loss/' + k])) <= 1e-4 * abs(float(g['loss/' + k])), (k, float(losses[k]), float(g['loss/' + k]))
    opt = make_optimizer(model, cfg)
    opt.zero_grad()
    losses['loss'].backward()
    params = dict(model.named_parameters())
    names, norms = [str(n) for n in g['grad/names']], g['grad/norms']
    assert {n for n, p in params.items() if p.grad is not None} == set(names)
    total = 0.0
    big = float(norms.max())
    for n, want_norm in zip(names, norms):
        gn = float(params[n].grad.double().norm())
        total += gn * gn
        # 1e-3 relative for every parameter whose gradient matters, absolute floor for the tiny ones
        assert abs(gn - want_norm) <= grad_tol * want_norm + 1e-5 * big, (n, gn, float(want_norm))
    assert abs(total ** 0.5 - float(g['grad/total_norm'])) <= grad_tol * float(g['grad/total_norm'])
    for key in g.files:
        if key.startswith('grad/full/'):
            assert_close(params[key[len('grad/full/'):]].grad.cpu(), g[key], full_tol, key)
    # a strided slice of EVERY gradient (up to 64 entries each), within full_tol of the gradient's largest entry: the norms alone would let a
    # wrong gradient of the right size through
    offs, absmax = g['grad/slice_offsets'], g['grad/absmax']
    for i, n in enumerate(names):
        flat = params[n].grad.reshape(-1)
        got = flat[::max(1, flat.numel() // 64)][:64].cpu().double().numpy()
        want_sl = g['grad/slices'][offs[i]:offs[i + 1]].astype('float64')
        assert got.shape == want_sl.shape, n
        assert float(abs(got - want_sl).max()) <= full_tol * float(absmax[i]) + 1e-6 * big, (n, float(abs(got - want_sl).max()), float(absmax[i]))
    opt.step()
    assert_close(model.transformer.out_proj.weight.detach().cpu(), g['after_step/out_proj_weight'], 1e-4, 'out_proj.weight after Adam')
    psum = sum(float(p.detach().double().sum()) for p in model.parameters())
    assert abs(psum - float(g['after_step/param_sum'])) <= 1e-4 * abs(float(g['after_step/param_sum'])) + 1e-2


@pytest.mark.parametrize('variant,fixture', [('micro_e', 'train_micro_se3ete.npz'), ('micro_i', 'train_micro_se3eti.npz')])
def test_training_step_matches_reference(golden_dir, variant, fixture):
    g, cfg, model, dd = _setup(variant, fixture, golden_dir)
    _check_training_step(g, cfg, model, dd)


def _setup_fullsize(golden_dir):
    """BASELINE.json configs[4] at its own size: SE3ET-E (17.1 M parameters), pair 0 of the bench workload (5000 + 5000 points), name-keyed
    synthetic weights (seed 7) -- tests/golden/train_c2_se3ete_5k.npz holds the genuine reference's forward + OverallLoss + backward + Adam
    step on exactly that (generate_golden.py train_fullsize)."""
    from se3et_amd.data import registration_collate_fn_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    g = np.load(golden_dir + '/train_c2_se3ete_5k.npz')
    cfg = make_cfg('se3ete')
    model = load_synthetic_weights(create_model(cfg), seed=7).cuda().train()
    d = dict(ref_points=g['ref'], src_points=g['src'], ref_feats=np.ones((len(g['ref']), 1), np.float32),
             src_feats=np.ones((len(g['src']), 1), np.float32), transform=g['transform'])
    b = cfg.backbone
    dd = registration_collate_fn_stack_mode([d], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    return g, cfg, model, dd


def test_fullsize_training_step_matches_reference(golden_dir):
    """Losses 1e-4, the gradient norm of every one of the 346 parameters 1e-3, seven complete gradients 2e-3, parameters after the Adam step:
    the hand-written backward kernels (KPConv at 80 000 / 51 742 / 21 411 / 5 506 points up to 256 channels, GroupNorm, embedding at 382
    superpoints, Sinkhorn, LayerNorm) against the reference's autograd at the size they run at."""
    g, cfg, model, dd = _setup_fullsize(golden_dir)
    _check_training_step(g, cfg, model, dd)


def test_ddp_wrapped_model_gives_the_unwrapped_gradients(golden_dir):
    """The real SE3ET-E in DistributedDataParallel (backend nccl = RCCL, world size 1, this process): custom autograd Functions,
    find_unused_parameters, float-atomic scatter -- gradients, reduced by DDP's bucketed all-reduce, equal the unwrapped step's."""
    import torch.distributed as dist
    from se3et_amd.training import OverallLoss, distributed_model
    g, cfg, model, dd = _setup_fullsize(golden_dir)
    targets = (torch.from_numpy(g['target/ref']).long(), torch.from_numpy(g['target/src']).long())
    loss_fn = OverallLoss(cfg)

    def grads(net):
        for p in model.parameters():
            p.grad = None
        out = net(dd, train=True, targets=targets)
        loss = loss_fn(out, dd)['loss']
        loss.backward()
        return float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    loss0, g0 = grads(model)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29541', rank=0, world_size=1)
    try:
        assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
        net = distributed_model(model, torch.device('cuda', torch.cuda.current_device()))
        loss1, g1 = grads(net)
        loss2, g2 = grads(net)              # a second step through the same wrapper (bucket rebuild after the first)
    finally:
        if created:
            dist.destroy_process_group()
    assert abs(loss1 - loss0) <= 1e-5 * abs(loss0) and abs(loss2 - loss0) <= 1e-5 * abs(loss0)
    assert set(g0) == set(g1) == set(g2)
    big = max(float(v.norm()) for v in g0.values())
    for n in g0:
        # float atomics in the KPConv scatter: arrival-order round-off between two runs, nothing more
        for other in (g1[n], g2[n]):
            assert float((other - g0[n]).norm()) <= 2e-4 * float(g0[n].norm()) + 1e-6 * big, n


def test_restatements_match_the_kernels():
    """Forward values of the PyTorch restatements used for backward (se3et_amd/autograd.py) against the kernels they stand in for."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    from se3et_amd import ops, tables
    g = torch.Generator().manual_seed(21)
    dev = 'cuda'
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    # KPConv
    P, Ns, NN, Cin, Cout = 300, 400, 20, 16, 16
    s_pts = (torch.rand(Ns, 3, generator=g) * 0.3).to(dev)
    q_pts = s_pts[:P].contiguous()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(NN, dim=1, largest=False)[1]
    idx[d.gather(1, idx) > 0.0625 ** 2] = Ns
    x, w = rn(Ns, 6, Cin), rn(6, 6, Cin, Cout) / (36 * Cin) ** 0.5
    kp = torch.from_numpy(tables.kernel_points(0.0625)).to(dev)
    kidx, ridx = torch.from_numpy(tables.kernel_slot_table()).to(dev), torch.from_numpy(tables.anchor_slot_table()).to(dev)
    assert_close(AG.kpconv_inter_so3(x, q_pts, s_pts, idx, kp, w, kidx, ridx, 0.05), SF.kpconv_inter_so3(x, q_pts, s_pts, idx, kp, w, kidx, ridx, 0.05), 1e-5, 'kpconv')
    # GroupNorm (+ bias, residual, LeakyReLU), max-pool, add + LayerNorm
    y, gw, gb, res, xb = rn(P, 6, 32), rn(32), rn(32), rn(P, 6, 32), rn(32)
    assert_close(AG.group_norm_rows(y, gw, gb, res, xb, 4, 1e-5, 0.1, None), SF.group_norm_rows(y, gw, gb, 4, 1e-5, 0.1, res, xb), 1e-5, 'group norm')
    assert_close(AG.neighbor_max_pool(x, idx), SF.neighbor_max_pool(x, idx), 1e-6, 'max pool')
    h, r = rn(6, 77, 32), rn(77, 32)
    assert_close(AG.add_layer_norm(h, r, gw, gb, xb, 1e-5), SF.add_layer_norm(h, r, gw, gb, 1e-5, xb), 1e-5, 'add + layer norm')
    # embeddings and attention
    N, C, H = 61, 32, 4
    pts = (torch.rand(N, 3, generator=g)).to(dev)
    div = torch.exp(torch.arange(0, C, 2).float() * (-np.log(10000.0) / C)).to(dev)
    wd, bd, wa, ba = rn(C, C) / C ** 0.5, rn(C) * 0.1, rn(C, C) / C ** 0.5, rn(C) * 0.1
    w1 = torch.from_numpy(tables.wigner_tables()[1]).to(dev)
    knn = ops.knn3_stack(pts, [N])
    emb, eq = SF.geometric_embedding(pts, div, wd, bd, wa, ba, 0.2, 15.0, 3, wigner_d1=w1, knn=knn)
    off = ~torch.eye(N, dtype=torch.bool, device=dev)
    assert_close(AG.geometric_embedding(pts, div, wd, bd, wa, ba, knn, 0.2, 15.0)[off], emb[off], 1e-4, 'geometric embedding')
    q, k, v, wp, we = rn(6, N, C), rn(6, N, C), rn(6, N, C), rn(C, C) / C ** 0.5, rn(C, 4) * 0.5
    vt = SF.project_values_transposed(v, torch.eye(C, device=dev), torch.zeros(C, device=dev))
    assert_close(AG.rpe_attention(q, k, vt, emb, wp, eq, we, H), SF.rpe_attention(q, k, vt, emb, wp, eq, we, H)[0], 1e-4, 'rpe attention (eq)')
    assert_close(AG.rpe_attention(q[0], k[0], vt[0], emb, wp, None, None, H), SF.rpe_attention(q[0], k[0], vt[0], emb, wp, None, None, H)[0], 1e-4, 'rpe attention')
    M = 53
    k2, v2 = rn(6, M, C), rn(6, M, C)
    vt2 = SF.project_values_transposed(v2, torch.eye(C, device=dev), torch.zeros(C, device=dev))
    assert_close(AG.cross_attention(q[0], k2[0], vt2[0], H), SF.cross_attention(q[0], k2[0], vt2[0], H), 1e-4, 'cross attention')
    assert_close(AG.cross_attention(q[0], k2[0], vt2, H), SF.cross_attention(q[0], k2[0], vt2, H), 1e-4, 'cross attention, anchor values')
    trace = torch.from_numpy(tables.trace_indices()[0]).to(dev)
    for mode in ('a_soft', 'r_soft'):
        got, want = AG.cross_attention_eq(q, k2, vt2, trace, H, mode), SF.cross_attention_eq(q, k2, vt2, H, mode, trace)
        for a, b_, name in zip(got, want, ('hidden', 'weights', 'mix')):
            assert_close(a, b_, 1e-4, 'cross attention eq %s %s' % (mode, name))
    # Sinkhorn
    sc = rn(9, 20, 24)
    rm, cm = torch.rand(9, 20, generator=g).to(dev) > 0.2, torch.rand(9, 24, generator=g).to(dev) > 0.2
    al = torch.tensor(0.7, device=dev)
    got, want = AG.log_optimal_transport(sc, al, rm, cm, 100, 1e12), SF.log_optimal_transport(sc, rm, cm, al, 100, 1e12)
    valid = want > -1e11
    assert torch.equal(got > -1e11, valid)
    assert float((got[valid] - want[valid]).abs().max()) <= 1e-4 * float(want[valid].abs().max())


def test_gradients_of_one_wrapped_op_against_finite_differences():
    """The kernel-forward / restatement-backward wrapper on one op (KPConv): directional derivative by central differences of the
    KERNEL forward against the autograd gradient (the two sides of `differentiable` describe the same function)."""
    from se3et_amd import functional as SF
    from se3et_amd import tables
    g = torch.Generator().manual_seed(22)
    P, NN, Cin, Cout = 120, 12, 8, 16
    pts = (torch.rand(P, 3, generator=g) * 0.2).cuda()
    idx = ((pts[:, None] - pts[None]) ** 2).sum(-1).topk(NN, dim=1, largest=False)[1]
    x = torch.randn(P, 6, Cin, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(6, 6, Cin, Cout, generator=g) / (36 * Cin) ** 0.5).cuda().requires_grad_(True)
    kp = torch.from_numpy(tables.kernel_points(0.0625)).cuda()
    kidx, ridx = torch.from_numpy(tables.kernel_slot_table()).cuda(), torch.from_numpy(tables.anchor_slot_table()).cuda()
    f = lambda x_, w_: SF.kpconv_inter_so3(x_, pts, pts, idx, kp, w_, kidx, ridx, 0.05)
    c = torch.randn(P, 6, Cout, generator=g).cuda()
    (f(x, w) * c).sum().backward()
    dx, dw = torch.randn(x.shape, generator=g).cuda(), torch.randn(w.shape, generator=g).cuda()
    eps = 1e-2
    with torch.no_grad():
        fd = ((f(x + eps * dx, w + eps * dw) - f(x - eps * dx, w - eps * dw)) * c).sum() / (2 * eps)
    an = (x.grad * dx).sum() + (w.grad * dw).sum()
    assert abs(float(fd) - float(an)) <= 2e-3 * abs(float(an)), (float(fd), float(an))


@pytest.mark.parametrize('P,Ns,NN,Cin,Cout,strided', [(300, 400, 20, 16, 16, True), (257, 257, 33, 32, 64, False), (130, 130, 38, 24, 40, False)])
def test_hand_written_backward_kernels_match_the_restatement_gradients(P, Ns, NN, Cin, Cout, strided):
    """KPConv (csrc/kpconv_so3.hip: kpconv_scatter_kernel + two GEMMs), neighbour max-pool and padded row gather: gradients of the
    hand-written backward against autograd through the PyTorch restatements (se3et_amd/autograd.py), padded neighbour tables included."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    from se3et_amd import tables
    g = torch.Generator().manual_seed(31 + P)
    dev = 'cuda'
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    s_pts = (torch.rand(Ns, 3, generator=g) * 0.3).to(dev)
    q_pts = s_pts[:P].contiguous()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(NN, dim=1, largest=False)[1]
    idx[d.gather(1, idx) > 0.0625 ** 2] = Ns                       # padded entries
    kp = torch.from_numpy(tables.kernel_points(0.0625)).to(dev)
    kidx, ridx = torch.from_numpy(tables.kernel_slot_table()).to(dev), torch.from_numpy(tables.anchor_slot_table()).to(dev)
    x0, w0 = rn(Ns, 6, Cin), rn(6, 6, Cin, Cout) / (36 * Cin) ** 0.5
    c = rn(P, 6, Cout)
    grads = []
    for fn in (SF.kpconv_inter_so3, AG.kpconv_inter_so3):
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        (fn(x, q_pts, s_pts, idx, kp, w, kidx, ridx, 0.05) * c).sum().backward()
        grads.append((x.grad, w.grad))
    assert_close(grads[0][0], grads[1][0], 2e-5, 'kpconv dL/dx')
    assert_close(grads[0][1], grads[1][1], 2e-5, 'kpconv dL/dW')
    # only one of the two gradients requested
    x = x0.clone().requires_grad_(True)
    (SF.kpconv_inter_so3(x, q_pts, s_pts, idx, kp, w0, kidx, ridx, 0.05) * c).sum().backward()
    assert_close(x.grad, grads[1][0], 2e-5, 'kpconv dL/dx alone')
    w = w0.clone().requires_grad_(True)
    (SF.kpconv_inter_so3(x0, q_pts, s_pts, idx, kp, w, kidx, ridx, 0.05) * c).sum().backward()
    assert_close(w.grad, grads[1][1], 2e-5, 'kpconv dL/dW alone')
    # max-pool and padded gather
    c2 = rn(P, 6, Cin)
    for name, hip, ref in (('max pool', SF.neighbor_max_pool, AG.neighbor_max_pool),):
        got = []
        for fn in (hip, ref):
            x = x0.clone().requires_grad_(True)
            (fn(x, idx) * c2).sum().backward()
            got.append(x.grad)
        assert_close(got[0], got[1], 1e-6, name + ' dL/dx')
    idx1 = idx[:, 0].clone()
    idx1[::7] = Ns
    got = []
    for fn in (SF.gather_rows_padded, AG.gather_rows_padded):
        x = x0.clone().requires_grad_(True)
        (fn(x, idx1) * c2).sum().backward()
        got.append(x.grad)
    assert_close(got[0], got[1], 1e-6, 'padded gather dL/dx')


@pytest.mark.parametrize('rows,C,groups,slope,with_res,with_xb,segments', [
    (1800, 32, 4, 0.1, True, True, None), (1501, 64, 32, None, False, False, None), (2400, 48, 8, 0.1, True, False, [0, 700, 1500, 2400]),
    (600, 16, 8, 0.1, False, True, [0, 10, 600])])
def test_group_norm_backward_kernels_match_the_restatement_gradients(rows, C, groups, slope, with_res, with_xb, segments):
    """csrc/rowops.hip gn_bwd_*: all five gradients (x, weight, bias, residual, producing layer's bias) against autograd through the
    PyTorch restatement, several independently normalised row segments included."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(rows)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    base = [rn(rows // 6, 6, C) if rows % 6 == 0 else rn(rows, C), rn(C), rn(C), rn(rows // 6, 6, C) if rows % 6 == 0 else rn(rows, C), rn(C)]
    if not with_res: base[3] = None
    if not with_xb: base[4] = None
    c = torch.randn(base[0].shape, generator=g).cuda()
    grads = []
    for hip in (True, False):
        t = [b.clone().requires_grad_(True) if b is not None else None for b in base]
        if hip:
            y = SF.group_norm_rows(t[0], t[1], t[2], groups, 1e-5, slope, t[3], t[4], segments)
        else:
            y = AG.group_norm_rows(t[0], t[1], t[2], t[3], t[4], groups, 1e-5, slope, segments)
        (y * c).sum().backward()
        grads.append([b.grad if b is not None else None for b in t])
    for name, a, b in zip(('x', 'weight', 'bias', 'residual', 'x_bias'), grads[0], grads[1]):
        if b is not None:
            assert_close(a, b, 2e-5, 'group norm d/d' + name)


@pytest.mark.parametrize('B,R,C,iters', [(9, 20, 24, 100), (5, 64, 64, 100), (3, 64, 37, 30), (2, 100, 90, 50)])
def test_sinkhorn_backward_kernel_matches_the_restatement_gradients(B, R, C, iters):
    """csrc/sinkhorn.hip sinkhorn_bwd_kernel against autograd through the PyTorch restatement of the 2 x iters logsumexp passes: gradients
    of the scores (valid entries) and of the dustbin score alpha, random masks, a loss that reads the dustbin row / column too."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(B * 100 + R)
    sc0 = torch.randn(B, R, C, generator=g).cuda()
    rm, cm = (torch.rand(B, R, generator=g) > 0.2).cuda(), (torch.rand(B, C, generator=g) > 0.2).cuda()
    rm[0], cm[0] = True, True
    c = torch.randn(B, R + 1, C + 1, generator=g).cuda()
    valid = torch.ones(B, R + 1, C + 1, dtype=torch.bool, device='cuda')
    valid[:, :R] &= rm[:, :, None]
    valid[:, :, :C] &= cm[:, None, :]
    grads = []
    for hip in (True, False):
        sc, al = sc0.clone().requires_grad_(True), torch.tensor(0.7, device='cuda', requires_grad=True)
        out = SF.log_optimal_transport(sc, rm, cm, al, iters, 1e12) if hip else AG.log_optimal_transport(sc, al, rm, cm, iters, 1e12)
        (torch.where(valid, out, torch.zeros_like(out)) * c).sum().backward()
        grads.append((sc.grad, al.grad))
    m = valid[:, :R, :C]
    assert float(grads[0][0][~m].abs().max()) == 0.0 if bool((~m).any()) else True
    assert_close(grads[0][0][m], grads[1][0][m], 1e-4, 'sinkhorn d/dscores')
    assert abs(float(grads[0][1]) - float(grads[1][1])) <= 1e-4 * max(1.0, abs(float(grads[1][1]))), (float(grads[0][1]), float(grads[1][1]))


@pytest.mark.parametrize('N,C,eq', [(61, 32, True), (120, 128, False), (17, 64, True)])
def test_embedding_backward_kernel_matches_the_restatement_gradients(N, C, eq):
    """csrc/geo_embedding.hip geo_embedding_bwd_operands_kernel + two GEMMs against autograd through the PyTorch restatement: gradients of
    proj_d / proj_a weights and biases.  The cotangent is zero on the n == m diagonal (its distance index is the square root of a rounding
    residue in both implementations: noise of 1e-3 index units); the angle terms are compared in the Frobenius norm, because the two
    implementations may pick different winners of `max` where two angle responses agree to round-off (a handful of (pair, channel)
    elements out of millions, each moving single entries of the gradient)."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    from se3et_amd import ops, tables
    g = torch.Generator().manual_seed(N)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    pts = torch.rand(N, 3, generator=g).cuda()
    div = torch.exp(torch.arange(0, C, 2).float() * (-np.log(10000.0) / C)).cuda()
    base = [rn(C, C) / C ** 0.5, rn(C) * 0.1, rn(C, C) / C ** 0.5, rn(C) * 0.1]
    w1 = torch.from_numpy(tables.wigner_tables()[1]).cuda() if eq else None
    knn = ops.knn3_stack(pts, [N])
    c = rn(N, N, C) * (~torch.eye(N, dtype=torch.bool, device='cuda'))[:, :, None]
    grads = []
    for hip in (True, False):
        t = [b.clone().requires_grad_(True) for b in base]
        if hip:
            out = SF.geometric_embedding(pts, div, t[0], t[1], t[2], t[3], 0.2, 15.0, 3, wigner_d1=w1, knn=knn)
            emb = out[0] if eq else out
        else:
            emb = AG.geometric_embedding(pts, div, t[0], t[1], t[2], t[3], knn, 0.2, 15.0)
        (emb * c).sum().backward()
        grads.append([b.grad for b in t])
    assert_close(grads[0][0], grads[1][0], 1e-4, 'embedding d/dproj_d.weight')
    assert_close(grads[0][1], grads[1][1], 1e-5, 'embedding d/dproj_d.bias')
    assert_close(grads[0][3], grads[1][3], 1e-5, 'embedding d/dproj_a.bias')
    fro = float((grads[0][2] - grads[1][2]).norm() / grads[1][2].norm())
    assert fro <= 2e-4, 'embedding d/dproj_a.weight: relative Frobenius error %.2e' % fro


@pytest.mark.parametrize('shape,res_shape,C,with_hb', [((6, 77), (77,), 32, True), ((300,), (300,), 256, False), ((6, 41), (6, 41), 512, True)])
def test_layer_norm_backward_kernel_matches_the_restatement_gradients(shape, res_shape, C, with_hb):
    """csrc/rowops.hip add_ln_bwd_kernel: the five gradients of LayerNorm(hidden [+ bias] + residual), the residual broadcast over the
    anchor axis included, against autograd through the PyTorch restatement."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(C + len(shape))
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    base = [rn(*shape, C), rn(*res_shape, C), rn(C), rn(C), rn(C) if with_hb else None]
    c = rn(*shape, C)
    grads = []
    for hip in (True, False):
        t = [b.clone().requires_grad_(True) if b is not None else None for b in base]
        y = SF.add_layer_norm(t[0], t[1], t[2], t[3], 1e-5, t[4]) if hip else AG.add_layer_norm(t[0], t[1], t[2], t[3], t[4], 1e-5)
        (y * c).sum().backward()
        grads.append([b.grad if b is not None else None for b in t])
    for name, a, b in zip(('hidden', 'residual', 'weight', 'bias', 'hidden_bias'), grads[0], grads[1]):
        if b is not None:
            assert_close(a, b, 2e-5, 'layer norm d/d' + name)


def test_kpconv_kernels_with_other_slot_tables_match_the_restatement():
    """Slot tables that are not the compiled-in SE3ET ones take the table-driven variants of the gather / scatter kernels (kidx / ridx as
    kernel arguments) and the library-GEMM path: forward and both gradients against the PyTorch restatement."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    from se3et_amd import tables
    g = torch.Generator().manual_seed(77)
    dev = 'cuda'
    P, Ns, NN, Cin, Cout = 150, 220, 24, 16, 32
    s_pts = (torch.rand(Ns, 3, generator=g) * 0.3).to(dev)
    q_pts = s_pts[:P].contiguous()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(NN, dim=1, largest=False)[1]
    idx[d.gather(1, idx) > 0.0625 ** 2] = Ns
    kp = torch.from_numpy(tables.kernel_points(0.0625)).to(dev)
    kidx = torch.from_numpy(tables.kernel_slot_table()).to(dev)[:, [1, 2, 3, 4, 5, 0]].contiguous()      # a relabelling of the output anchors
    ridx = torch.from_numpy(tables.anchor_slot_table()).to(dev)[:, [1, 2, 3, 4, 5, 0]].contiguous()
    x0, w0 = torch.randn(Ns, 6, Cin, generator=g).to(dev), (torch.randn(6, 6, Cin, Cout, generator=g) / (36 * Cin) ** 0.5).to(dev)
    c = torch.randn(P, 6, Cout, generator=g).to(dev)
    res = []
    for fn in (SF.kpconv_inter_so3, AG.kpconv_inter_so3):
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        y = fn(x, q_pts, s_pts, idx, kp, w, kidx, ridx, 0.05)
        (y * c).sum().backward()
        res.append((y.detach(), x.grad, w.grad))
    for name, a, b in zip(('forward', 'dL/dx', 'dL/dW'), res[0], res[1]):
        assert_close(a, b, 2e-5, 'kpconv with relabelled tables: ' + name)


@pytest.mark.parametrize('anchors,use_eq', [(None, False), (6, True), (6, False)])
def test_attention_backward_matches_autograd(anchors, use_eq):
    """The hand-derived backward of rpe_attention / cross_attention (se3et_amd/attention_bwd.py: logits recomputed by the forward's HIP
    kernel, batched GEMMs) against reverse-mode differentiation of the PyTorch restatements, at N = 382, C = 256 (the coarse level of a
    5k-point cloud): every input gradient within 1e-3 of the largest entry."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    g = torch.Generator().manual_seed(5)
    N, M, C, H = 382, 382, 256, 4
    lead = () if anchors is None else (anchors,)
    Mp = (M + 31) // 32 * 32
    mk = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).cuda().requires_grad_(True)
    q, k = mk(*lead, N, C), mk(*lead, M, C)
    vt = mk(*lead, C, Mp)
    emb = mk(N, M, C, scale=0.5)
    w_p = mk(C, C, scale=C ** -0.5)
    eq_emb = (torch.randn(anchors, N, M, 4, generator=g)).cuda() if use_eq else None
    w_eq = mk(C, 4, scale=0.5) if use_eq else None
    go = torch.randn(*lead, N, C, generator=g).cuda()
    leaves = [t for t in (q, k, vt, emb, w_p, w_eq) if t is not None]
    got = torch.autograd.grad(SF.rpe_attention(q, k, vt, emb, w_p, eq_emb, w_eq, H)[0], leaves, go)
    want = torch.autograd.grad(AG.rpe_attention(q, k, vt, emb, w_p, eq_emb, w_eq, H), leaves, go)
    for name, a, b in zip(('q', 'k', 'vt', 'emb', 'w_p', 'w_eq'), got, want):
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()), name
    assert float(got[2][..., M:].abs().max()) == 0.0             # the padded key columns receive no gradient
    if anchors is None or not use_eq:
        # plain cross attention: shared scores, values per anchor or not
        q2, k2 = mk(N, C), mk(M - 40, C)
        vt2 = mk(*lead, C, Mp)
        go2 = torch.randn(*lead, N, C, generator=g).cuda()
        got = torch.autograd.grad(SF.cross_attention(q2, k2, vt2, H), (q2, k2, vt2), go2)
        want = torch.autograd.grad(AG.cross_attention(q2, k2, vt2, H), (q2, k2, vt2), go2)
        for name, a, b in zip(('q', 'k', 'vt'), got, want):
            assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()), 'cross ' + name


@pytest.mark.parametrize('mode', ['a_soft', 'r_soft'])
def test_equivariant_cross_attention_backward_matches_autograd(mode):
    """se3et_amd/attention_bwd.py::_CrossAttentionEq (all three outputs differentiable) against reverse-mode differentiation of the
    restatement, with gradients arriving on the hidden states, the weights and the mixing matrix."""
    from se3et_amd import autograd as AG
    from se3et_amd import functional as SF
    from se3et_amd import tables
    g = torch.Generator().manual_seed(9)
    A, N, M, C, H = 6, 382, 350, 256, 4
    Mp = (M + 31) // 32 * 32
    mk = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).cuda().requires_grad_(True)
    q, k, vt = mk(A, N, C, scale=0.7), mk(A, M, C, scale=0.7), mk(A, C, Mp)
    trace = torch.from_numpy(tables.trace_indices()[0]).long().cuda()
    outs_a = SF.cross_attention_eq(q, k, vt, H, mode, trace)
    outs_b = AG.cross_attention_eq(q, k, vt, trace, H, mode)
    gos = [torch.randn(o.shape, generator=g).cuda() for o in outs_b]
    for o_a, o_b in zip(outs_a, outs_b):
        assert float((o_a - o_b).abs().max()) <= 1e-4 * float(o_b.abs().max())
    got = torch.autograd.grad([o.reshape(b.shape) for o, b in zip(outs_a, outs_b)], (q, k, vt), gos)
    want = torch.autograd.grad(outs_b, (q, k, vt), gos)
    for name, a, b in zip(('q', 'k', 'vt'), got, want):
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()), name


def test_kpconv_backward_is_bit_identical_between_runs():
    """csrc/kpconv_so3.hip: the gradient with respect to the input features is a scatter into support rows that many queries share.  With the
    64-bit fixed-point sums (the default: ops.KPCONV_BACKWARD_DETERMINISTIC) the result does not depend on the arrival order of the atomics:
    repeated runs are bit-identical, and they agree with the float-atomic form to float32 round-off at magnitudes from 1e-8 to 1e6."""
    from se3et_amd import ops, tables
    g = torch.Generator().manual_seed(21)
    Ns, P, NN, Cin, Cout, radius, sigma = 1500, 1100, 38, 32, 64, 0.0625, 0.05
    s_pts = (torch.rand(Ns, 3, generator=g) * 0.22).cuda()
    q_pts = s_pts[torch.randperm(Ns, generator=g)[:P]].contiguous()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(NN, dim=1, largest=False)[1]
    idx[d.gather(1, idx) > radius ** 2] = Ns
    x = torch.randn(Ns, 6, Cin, generator=g).cuda()
    w = (torch.randn(6, 6, Cin, Cout, generator=g) / (36 * Cin) ** 0.5).cuda()
    kp = torch.from_numpy(tables.kernel_points(radius)).cuda()
    kidx, ridx = torch.from_numpy(tables.kernel_slot_table()).cuda(), torch.from_numpy(tables.anchor_slot_table()).cuda()
    saved = ops.KPCONV_BACKWARD_DETERMINISTIC
    try:
        for scale in (1e-8, 1.0, 1e6):
            go = (torch.randn(P, 6, Cout, generator=g) * scale).cuda()
            ops.KPCONV_BACKWARD_DETERMINISTIC = True
            runs = [ops.kpconv_inter_so3_bwd(go, x, q_pts, s_pts, idx, kp, w, kidx, ridx, sigma, need_x=True, need_w=False)[0] for _ in range(4)]
            assert all(torch.equal(r, runs[0]) for r in runs[1:]), scale
            ops.KPCONV_BACKWARD_DETERMINISTIC = False
            flt = ops.kpconv_inter_so3_bwd(go, x, q_pts, s_pts, idx, kp, w, kidx, ridx, sigma, need_x=True, need_w=False)[0]
            assert float((runs[0] - flt).abs().max()) <= 2e-6 * float(flt.abs().max()), scale
            assert torch.equal(runs[0] == 0, flt == 0)                  # (support rows nobody gathers stay exactly zero)
    finally:
        ops.KPCONV_BACKWARD_DETERMINISTIC = saved


def test_fixed_point_scatters_propagate_non_finite_gradients_and_take_empty_inputs():
    """the order-independent (64-bit fixed-point) scatters of the default training backward quantise through an integer
    conversion -- a NaN / Inf gradient must not come out as a finite number: the bound word of the call is non-finite then and
    se3_fixed_to_float writes NaN.  Empty gradients / empty index tables return zeros like the float forms."""
    from se3et_amd import ops, tables
    g = torch.Generator().manual_seed(5)
    x = torch.randn(50, 6, 16, generator=g).cuda()
    idx = torch.randint(0, 51, (30, 9), generator=g).cuda()
    go = torch.randn(30, 6, 16, generator=g).cuda()
    clean = ops.neighbor_max_pool_bwd(x, idx, go)
    assert bool(torch.isfinite(clean).all())
    for bad in (float('nan'), float('inf'), float('-inf')):
        gb = go.clone()
        gb[4, 2, 3] = bad
        assert bool(torch.isnan(ops.neighbor_max_pool_bwd(x, idx, gb)).all()), bad
        rows = torch.randn(30, 9, 8, generator=g).cuda()
        rows[1, 1, 1] = bad
        assert bool(torch.isnan(ops.scatter_add_rows(rows, idx, 50)).all()), bad
    assert torch.equal(ops.neighbor_max_pool_bwd(x, idx[:0], go[:0]), torch.zeros_like(x))
    assert torch.equal(ops.scatter_add_rows(go[:0], idx[:0, 0], 50), torch.zeros(50, 6, 16, device='cuda'))
    # KPConv: a NaN in grad_out reaches dx as NaN; an empty query set gives zeros
    Ns, P, NN, Cin, Cout, radius, sigma = 300, 200, 20, 16, 32, 0.0625, 0.05
    s_pts = (torch.rand(Ns, 3, generator=g) * 0.2).cuda()
    q_pts = s_pts[:P].contiguous()
    nb = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1).topk(NN, dim=1, largest=False)[1]
    xx = torch.randn(Ns, 6, Cin, generator=g).cuda()
    w = (torch.randn(6, 6, Cin, Cout, generator=g) / (36 * Cin) ** 0.5).cuda()
    kp = torch.from_numpy(tables.kernel_points(radius)).cuda()
    kidx, ridx = torch.from_numpy(tables.kernel_slot_table()).cuda(), torch.from_numpy(tables.anchor_slot_table()).cuda()
    d = torch.randn(P, 6, Cout, generator=g).cuda()
    assert bool(torch.isfinite(ops.kpconv_inter_so3_bwd(d, xx, q_pts, s_pts, nb, kp, w, kidx, ridx, sigma, need_x=True, need_w=False)[0]).all())
    d[7, 0, 0] = float('nan')
    assert bool(torch.isnan(ops.kpconv_inter_so3_bwd(d, xx, q_pts, s_pts, nb, kp, w, kidx, ridx, sigma, need_x=True, need_w=False)[0]).all())
    dx0 = ops.kpconv_inter_so3_bwd(d[:0], xx, q_pts[:0], s_pts, nb[:0], kp, w, kidx, ridx, sigma, need_x=True, need_w=False)[0]
    assert torch.equal(dx0, torch.zeros_like(xx))


@pytest.mark.parametrize('variant,preset', [('micro_e', 'micro'), ('se3ete', 'c1_2k')])
def test_training_step_is_bit_identical_between_runs(variant, preset):
    """VERDICT round 4 (missing 5 / next 7): 'two runs bit-identical'.  Every scatter-add of the step is order-independent -- the KPConv input
    gradient, the max-pool and row-gather backward as 64-bit fixed-point sums, the LayerNorm parameter sums as per-workgroup partials added in
    order -- and the library GEMMs are run-to-run deterministic (tools/r5/determinism_probe.py): with the same ground-truth selection, three
    forward + backward passes give the same loss and the same 300-odd gradients, bit for bit."""
    from se3et_amd.data import registration_collate_fn_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    from se3et_amd.training import OverallLoss
    cfg = make_cfg(variant)
    model = load_synthetic_weights(create_model(cfg), 7).cuda().train()
    ref, src, T = make_pair(preset)
    d = dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32), src_feats=np.ones((len(src), 1), np.float32), transform=T)
    dd = registration_collate_fn_stack_mode([d], cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius, cfg.neighbor_limits)
    loss_fn = OverallLoss(cfg)

    def grads():
        for p in model.parameters():
            p.grad = None
        out = model(dd, train=True, rng=np.random.default_rng(3))      # (the same random selection of ground-truth patch pairs every time)
        loss = loss_fn(out, dd)['loss']
        loss.backward()
        return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    l0, g0 = grads()
    assert len(g0) > 100 and np.isfinite(l0)
    for _ in range(2):
        l1, g1 = grads()
        assert l1 == l0 and set(g1) == set(g0)
        differ = [n for n in g0 if not torch.equal(g1[n], g0[n])]
        assert not differ, differ[:8]
