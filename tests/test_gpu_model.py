"""GPU parity of the whole SE3ET forward (HIP path through se3et_amd) against outputs captured from the genuine
reference (tests/golden/*.npz).  Tolerance: 1e-4 relative (max-norm), the figure BASELINE.json states for fp32."""
import numpy as np
import pytest
import torch

from helpers import assert_close, assert_neighbors_equal, assert_pairs_equal_up_to_ties

pytestmark = pytest.mark.gpu


def _run(variant, pair, state=None, synth_seed=None, attention_dtype='float32', packed=True):
    from se3et_amd.data import registration_collate_fn_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    cfg = make_cfg(variant, attention_dtype=attention_dtype)
    model = create_model(cfg)
    if state is not None:
        model.load_state_dict(state, strict=True)
    else:
        load_synthetic_weights(model, synth_seed)
    model = model.cuda().eval()
    model.packed_inference = packed      # one pair through the packed-row kernels (default) or through the per-module path
    ref, src, T = make_pair(pair)
    d = dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32),
             src_feats=np.ones((len(src), 1), np.float32), transform=T)
    dd = registration_collate_fn_stack_mode([d], cfg.backbone.num_stages, cfg.backbone.init_voxel_size,
                                            cfg.backbone.init_radius, cfg.neighbor_limits)
    return model, dd, model(dd)


def _check_outputs(out, g, full):
    fc = out['feats_c'].cpu()
    assert_close(fc if full else fc[:, :, :64], g['out/feats_c'], 1e-4, 'feats_c')
    ff = out['feats_f'].cpu()
    assert_close(ff if full else ff[::8], g['out/feats_f'], 1e-4, 'feats_f')
    assert_close(out['ref_feats_c'].cpu(), g['out/ref_feats_c'], 1e-4, 'ref_feats_c')
    assert_close(out['src_feats_c'].cpu(), g['out/src_feats_c'], 1e-4, 'src_feats_c')
    ri, si = torch.from_numpy(g['out/ref_node_corr_indices']).long(), torch.from_numpy(g['out/src_node_corr_indices']).long()
    # scores of the reference pairs recomputed from the reference features give the tie structure
    rf, sf = torch.from_numpy(g['out/ref_feats_c']), torch.from_numpy(g['out/src_feats_c'])
    s = torch.exp(-(2 - 2 * rf @ sf.t()).clamp(min=0))
    s = (s / s.sum(1, keepdim=True)) * (s / s.sum(0, keepdim=True))
    assert_pairs_equal_up_to_ties((out['ref_node_corr_indices'], out['src_node_corr_indices']), out['node_corr_scores'],
                                  (ri, si), s[ri, si], rtol=2e-4, context='node correspondences')
    same = (out['ref_node_corr_indices'].cpu()[:8] == ri[:8]).all() and (out['src_node_corr_indices'].cpu()[:8] == si[:8]).all()
    if same:
        got, want = out['matching_scores'][:8].cpu(), torch.from_numpy(g['out/matching_scores_head'])
        valid = want > -1e11
        assert torch.equal(got > -1e11, valid)
        assert float((got[valid] - want[valid]).abs().max()) <= 1e-4 * float(want[valid].abs().max())      # log-domain transport scores
    same_all = torch.equal(out['ref_node_corr_indices'].cpu(), ri) and torch.equal(out['src_node_corr_indices'].cpu(), si)
    cs_got, cs_want = out['corr_scores'].cpu().double(), torch.from_numpy(g['out/corr_scores']).double()
    if same_all and cs_got.shape == cs_want.shape and float((cs_got.sort()[0] - cs_want.sort()[0]).abs().max()) <= 1e-4 * float(cs_want.max()):
        # identical correspondence set: the transform is a smooth function of it (1e-4)
        assert_close(out['estimated_transform'].cpu(), g['out/estimated_transform'], 1e-4, 'estimated_transform')
    else:
        # a correspondence flipped at a threshold (mutual top-k / 0.05 confidence / top-256 cut-off): another set is fitted
        assert_close(out['estimated_transform'].cpu(), g['out/estimated_transform'], 2e-3, 'estimated_transform (correspondence set differs)')


@pytest.mark.parametrize('packed', [True, False])
@pytest.mark.parametrize('variant,fixture', [('micro_e', 'micro_se3ete.npz'), ('micro_i', 'micro_se3eti.npz')])
def test_micro_model_matches_reference(golden_dir, variant, fixture, packed):
    g = np.load(golden_dir + '/' + fixture)
    state = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    model, dd, out = _run(variant, 'micro', state=state, packed=packed)
    for i in range(4):
        assert dd['lengths'][i].tolist() == g['data/lengths_%d' % i].tolist()
        assert torch.equal(dd['points'][i].cpu(), torch.from_numpy(g['data/points_%d' % i]))
        assert_neighbors_equal(dd['neighbors'][i], g['data/neighbors_%d' % i], dd['points'][i], dd['points'][i])
    _check_outputs(out, g, full=True)
    for i in range(len(model.cfg.geotransformer.blocks)):
        key = 'op/layer_%d/out0' % i
        assert key in g.files


@pytest.mark.parametrize('packed', [True, False])
@pytest.mark.parametrize('variant,fixture', [('se3ete2', 'synthw_se3ete2.npz'), ('se3eti2', 'synthw_se3eti2.npz'),
                                             ('se3ete', 'synthw_se3ete.npz'), ('se3eti_kitti', 'synthw_se3eti_kitti.npz'),
                                             ('se3eti2', 'synthw_se3eti2_c1.npz')])      # the last one = BASELINE.json configs[0]: SE3ET-I2, 2k+2k pair
def test_real_width_model_matches_reference(golden_dir, variant, fixture, packed):
    g = np.load(golden_dir + '/' + fixture)
    model, dd, out = _run(variant, str(g['pair']), synth_seed=int(g['synth_seed']), packed=packed)
    assert np.array_equal(np.stack([l.numpy() for l in dd['lengths']]), g['lengths'])
    _check_outputs(out, g, full=False)


@pytest.mark.parametrize('variant,fixture', [('se3ete2', 'synthw_se3ete2.npz'), ('se3eti2', 'synthw_se3eti2.npz')])
def test_bf16_attention_model_stays_close_to_the_reference(golden_dir, variant, fixture):
    """BASELINE.json configs[2] ('bf16 attention'): with the geometric embedding stored in bf16 (2^-9 relative rounding per
    element, everything else f32) the outputs stay within 1e-2 (max-norm relative) of the genuine reference's f32 outputs and
    the estimated transform within 1e-2; the kernel itself is exact to 1e-4 on the rounded embedding (test_gpu_ops.py)."""
    g = np.load(golden_dir + '/' + fixture)
    model, dd, out = _run(variant, str(g['pair']), synth_seed=int(g['synth_seed']), attention_dtype='bfloat16')
    assert model.transformer.embedding.embedding_dtype == torch.bfloat16
    assert_close(out['ref_feats_c'].cpu(), g['out/ref_feats_c'], 1e-2, 'ref_feats_c (bf16 attention)')
    assert_close(out['src_feats_c'].cpu(), g['out/src_feats_c'], 1e-2, 'src_feats_c (bf16 attention)')
    assert_close(out['feats_f'].cpu()[::8], g['out/feats_f'], 1e-4, 'feats_f (backbone only: unaffected)')
    assert_close(out['estimated_transform'].cpu(), g['out/estimated_transform'], 1e-2, 'estimated_transform (bf16 attention)')


def test_bf16_attention_kitti_sized_pair():
    """configs[2] at its full size (SE3ET-I KITTI configuration, 20k + 20k points): bf16-attention forward vs the f32 forward of
    the same model, single pair and two pairs per forward."""
    from se3et_amd.batched import forward_pairs
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.synthetic import make_pair
    model, dd, out32 = _run('se3eti_kitti', 'c3_20k', synth_seed=0)
    emb = model.transformer.embedding
    emb.embedding_dtype = torch.bfloat16
    out16 = model(dd)
    assert_close(out16['ref_feats_c'], out32['ref_feats_c'], 1e-2, 'ref_feats_c (bf16 vs f32 attention)')
    assert_close(out16['src_feats_c'], out32['src_feats_c'], 1e-2, 'src_feats_c (bf16 vs f32 attention)')
    assert_close(out16['estimated_transform'], out32['estimated_transform'], 1e-2, 'estimated_transform (bf16 vs f32 attention)')
    clouds = []
    for j in range(2):
        ref, src, _ = make_pair('c3_20k', index=j)
        clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
    lens = torch.tensor([len(c) for c in clouds], dtype=torch.int64)
    b = model.cfg.backbone
    data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, model.cfg.neighbor_limits)
    data['features'] = torch.ones((pts.shape[0], 1), device='cuda')
    outs16 = forward_pairs(model, data)
    emb.embedding_dtype = torch.float32
    outs32 = forward_pairs(model, data)
    for o16, o32 in zip(outs16, outs32):
        assert_close(o16['ref_feats_c'], o32['ref_feats_c'], 1e-2, 'batched ref_feats_c (bf16 vs f32 attention)')
        assert_close(o16['estimated_transform'], o32['estimated_transform'], 1e-2, 'batched estimated_transform')


@pytest.mark.parametrize('variant,preset,num_pairs', [('micro_e', 'micro', 2), ('micro_i', 'micro', 3), ('se3ete', 'c1_2k', 2),
                                                      ('se3ete', 'c2_5k', 3), ('se3eti', 'c2_5k', 2), ('se3eti_kitti', 'c3_20k', 2),
                                                      ('se3ete', 'c2_5k', 16), ('se3eti', 'c1_2k', 16)])      # 16 pairs = SE3_MAX_BATCH clouds
def test_multi_pair_forward_equals_single_pair_forward(variant, preset, num_pairs):
    """se3et_amd.batched.forward_pairs (B pairs stacked through pyramid, backbone and transformer) against the single-pair
    forward of the same pairs: identical pyramids per pair, features to float32 round-off (the only arithmetic difference is
    the shape-dependent summation order inside library GEMMs and the GroupNorm partial sums)."""
    from se3et_amd.batched import forward_pairs
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    cfg = make_cfg(variant)
    model = load_synthetic_weights(create_model(cfg)).cuda().eval()
    model.packed_inference = False       # the single-pair side of the comparison runs the per-module path
    b = cfg.backbone
    clouds, singles = [], []
    for p in range(num_pairs):
        ref, src, _ = make_pair(preset, index=p)
        clouds += [ref, src]
        pts = torch.from_numpy(np.concatenate([ref, src], 0)).cuda()
        d = precompute_data_stack_mode(pts, torch.tensor([len(ref), len(src)]), b.num_stages, b.init_voxel_size, b.init_radius,
                                       cfg.neighbor_limits)
        d['features'] = torch.ones((pts.shape[0], 1), device='cuda')
        singles.append((d, model(d)))
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
    dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius,
                                    cfg.neighbor_limits)
    dd['features'] = torch.ones((pts.shape[0], 1), device='cuda')
    # the stacked pyramid restricted to a pair is the pair's own pyramid (points bit-exact, tables equal as neighbour sets)
    for s in range(b.num_stages):
        lens = dd['lengths'][s].tolist()
        start = 0
        for p, (d, _) in enumerate(singles):
            n = lens[2 * p] + lens[2 * p + 1]
            assert d['lengths'][s].tolist() == lens[2 * p:2 * p + 2]
            assert torch.equal(dd['points'][s][start:start + n], d['points'][s])
            got = dd['neighbors'][s][start:start + n]
            want = d['neighbors'][s]
            total, own = dd['points'][s].shape[0], d['points'][s].shape[0]
            g = torch.where((got < 0) | (got >= total), torch.full_like(got, own), got - start)[:, :want.shape[1]]
            assert_neighbors_equal(g.cpu(), want.cpu(), d['points'][s].cpu(), d['points'][s].cpu(), context='stage %d pair %d' % (s, p))
            assert bool(((dd['neighbors'][s][start:start + n][:, want.shape[1]:] < 0) |
                         (dd['neighbors'][s][start:start + n][:, want.shape[1]:] >= total)).all())
            start += n
    outs = forward_pairs(model, dd)
    assert len(outs) == num_pairs
    for p, ((_, want), got) in enumerate(zip(singles, outs)):
        for key in ('feats_c', 'feats_f', 'ref_feats_c', 'src_feats_c'):
            assert_close(got[key].cpu(), want[key].cpu(), 2e-5, 'pair %d %s' % (p, key))
        # the same superpoint pairs with the same scores; a pair may only be missing if its score sits at the top-k cut-off
        sw = dict(zip(zip(want['ref_node_corr_indices'].tolist(), want['src_node_corr_indices'].tolist()),
                      want['node_corr_scores'].tolist()))
        sg = dict(zip(zip(got['ref_node_corr_indices'].tolist(), got['src_node_corr_indices'].tolist()),
                      got['node_corr_scores'].tolist()))
        assert len(sw) == len(sg)
        cut = min(sw.values())
        for key in set(sw) | set(sg):
            if key in sw and key in sg:
                assert abs(sw[key] - sg[key]) <= 2e-4 * sw[key], 'pair %d: score of %s' % (p, key)
            else:
                assert abs(sw.get(key, sg.get(key)) - cut) <= 2e-4 * cut, 'pair %d: %s missing on one side' % (p, key)
        if torch.equal(got['ref_node_corr_indices'], want['ref_node_corr_indices']) and \
                torch.equal(got['src_node_corr_indices'], want['src_node_corr_indices']):
            valid = want['matching_scores'] > -1e11
            assert torch.equal(got['matching_scores'] > -1e11, valid)
            assert float((got['matching_scores'][valid] - want['matching_scores'][valid]).abs().max()) <= 1e-4 * float(want['matching_scores'][valid].abs().max())
            if got['ref_corr_points'].shape == want['ref_corr_points'].shape:
                assert_close(got['estimated_transform'].cpu(), want['estimated_transform'].cpu(), 2e-3, 'pair %d transform' % p)


@pytest.mark.parametrize('variant,preset,num_pairs', [('micro_e', 'micro', 2), ('micro_i', 'micro', 1), ('se3ete', 'c2_5k', 1), ('se3ete', 'c2_5k', 3),
                                                      ('se3eti', 'c2_5k', 2), ('se3eti_kitti', 'c3_20k', 2)])
def test_transformer_issued_from_c_equals_the_python_schedule(variant, preset, num_pairs):
    """csrc/transformer_driver.hip (se3_transformer_forward: the ten blocks and out_proj as ONE host call, ~130 launches issued from C) against
    se3et_amd.batched.transformer_pairs' Python schedule: the same kernels with the same operands in the same order, so the forward's outputs
    must be IDENTICAL -- SE3ET-E (self_eq / cross_a_soft / cross_r_soft + eq2inv + rotcompress / self / cross), SE3ET-I (anchor values
    through the plain cross blocks), the KITTI configuration, the micro models (head dimension 8: f32 attention kernels)."""
    from se3et_amd import cdriver
    from se3et_amd.batched import forward_pairs
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    cfg = make_cfg(variant)
    model = load_synthetic_weights(create_model(cfg)).cuda().eval()
    if variant == 'micro_e':              # 32 channels: the equivariant cross attention's Gram statistics kernel takes 128 / 256 -> Python schedule
        assert not cdriver.supported(model.transformer)
        return
    assert cdriver.supported(model.transformer)
    b = cfg.backbone
    clouds = []
    for p in range(num_pairs):
        ref, src, _ = make_pair(preset, index=p)
        clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()

    def run():
        dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius,
                                        cfg.neighbor_limits)
        dd['features'] = torch.ones((pts.shape[0], 1), device='cuda')
        return forward_pairs(model, dd)

    saved = cdriver.ENABLED
    try:
        cdriver.ENABLED = True
        got = run()
        got2 = run()                      # (cached plan, reused workspace)
        cdriver.ENABLED = False
        want = run()
    finally:
        cdriver.ENABLED = saved
    for p in range(num_pairs):
        for key in ('ref_feats_c', 'src_feats_c', 'ref_node_corr_indices', 'src_node_corr_indices', 'matching_scores', 'estimated_transform'):
            assert torch.equal(got[p][key], want[p][key]), (p, key, float((got[p][key].double() - want[p][key].double()).abs().max()))
            assert torch.equal(got2[p][key], want[p][key]), (p, key)


def test_c_issued_transformer_sees_weights_written_behind_the_version_counter():
    """cdriver's plan holds raw pointers to the f16 weight pieces.  After `p.data.copy_()` (no version bump) and the documented
    `ops.validate_weight_caches()` (or `clear_weight_caches()`), the C-issued transformer must multiply with the NEW weights: its outputs must
    equal the Python schedule's on the same weights, and differ from the outputs before the write."""
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

    def run():
        dd = precompute_data_stack_mode(pts, torch.tensor([len(ref), len(src)]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
        dd['features'] = torch.ones((pts.shape[0], 1), device='cuda')
        return forward_pairs(model, dd)[0]

    saved = cdriver.ENABLED
    try:
        cdriver.ENABLED = True
        before = run()
        for how in ('validate', 'clear'):
            lin = model.transformer.transformer.layers[4].output.expand           # an FFN layer issued from inside the C driver
            with torch.no_grad():
                lin.weight.data.copy_(lin.weight.data * 1.5)                      # behind the version counter
                model.transformer.out_proj.weight.data.copy_(model.transformer.out_proj.weight.data * 0.5)
            if how == 'validate':
                assert ops.validate_weight_caches() >= 2
            else:
                ops.clear_weight_caches()
            got = run()
            cdriver.ENABLED = False
            want = run()
            cdriver.ENABLED = True
            assert not torch.equal(got['ref_feats_c'], before['ref_feats_c'])
            for key in ('ref_feats_c', 'src_feats_c', 'matching_scores', 'estimated_transform'):
                assert torch.equal(got[key], want[key]), (how, key)
            before = got
    finally:
        cdriver.ENABLED = saved


@pytest.mark.parametrize('scale', [1e4, 1e-4])
def test_forward_with_scaled_input_features(scale):
    """un-normalised input features.  The features x s with the first layer's weights / s are the same network: the
    first KPConv (one input channel: slot sums + the dense kernel with per-row f16-split scales) sees magnitudes of 1e4 / 1e-4 where the
    bench sees 1, everything behind its GroupNorm is unchanged -- the outputs must stay at 1e-4 of the unscaled run, nothing may saturate."""
    from se3et_amd import ops
    from se3et_amd.data import registration_collate_fn_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    cfg = make_cfg('se3ete')
    model = load_synthetic_weights(create_model(cfg), 3).cuda().eval()
    ref, src, T = make_pair('c1_2k')

    def run(s):
        d = dict(ref_points=ref, src_points=src, ref_feats=np.full((len(ref), 1), s, np.float32),
                 src_feats=np.full((len(src), 1), s, np.float32), transform=T)
        dd = registration_collate_fn_stack_mode([d], cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius,
                                                cfg.neighbor_limits)
        with torch.no_grad():
            return model(dd)
    base = run(1.0)
    w = model.backbone.encoder1_1.interso3.conv.weights
    saved = w.detach().clone()
    ops.dense_saturated_rows(reset=True)
    try:
        with torch.no_grad():
            w.mul_(1.0 / scale)                     # (in place: the version counter moves, every weight cache follows)
        out = run(scale)
    finally:
        with torch.no_grad():
            w.copy_(saved)
    assert ops.dense_saturated_rows() == 0
    for key in ('feats_c', 'feats_f', 'ref_feats_c', 'src_feats_c'):
        assert_close(out[key].cpu(), base[key].cpu().numpy(), 1e-4, key + ' with features x %g' % scale)
    assert torch.equal(out['ref_node_corr_indices'], base['ref_node_corr_indices']) or \
        float((out['estimated_transform'] - base['estimated_transform']).abs().max()) <= 5e-3


@pytest.mark.parametrize('mode', ['policy', 'all'])
def test_union_kpconv_forward_matches_the_default_forward(mode):
    """The forward of a stacked batch with the union-staged KPConv selected (SE3_KPCONV_UNION=1: the four narrow layers; all: every layer,
    the wide ones as 128-column blocks, the coarse strided ones in several passes) against the default forward of the same batch: the same
    arithmetic up to the order of a neighbourhood's sum."""
    from se3et_amd import ops
    from se3et_amd.batched import forward_pairs
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    cfg = make_cfg('se3ete')
    model = load_synthetic_weights(create_model(cfg)).cuda().eval()
    b = cfg.backbone
    clouds = []
    for p in range(3):                                     # 30 000 stage-0 points: above ops.KPCONV_UNION_MIN_POINTS
        ref, src, _ = make_pair('c2_5k', index=p)
        clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()

    def run():
        dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius,
                                        cfg.neighbor_limits)
        dd['features'] = torch.ones((pts.shape[0], 1), device='cuda')
        return dd, forward_pairs(model, dd)
    saved = (ops.KPCONV_UNION, ops.KPCONV_UNION_ALL)
    try:
        ops.KPCONV_UNION = ops.KPCONV_UNION_ALL = False
        dd0, want = run()
        assert ops.point_order(dd0['points'][0]) is None
        ops.KPCONV_UNION, ops.KPCONV_UNION_ALL = True, mode == 'all'
        dd1, got = run()
        assert ops.point_order(dd1['points'][0]) is not None and (ops.point_order(dd1['points'][3]) is not None) == (mode == 'all')
    finally:
        ops.KPCONV_UNION, ops.KPCONV_UNION_ALL = saved
    for p, (w, g) in enumerate(zip(want, got)):
        for key in ('feats_c', 'feats_f', 'ref_feats_c', 'src_feats_c'):
            assert_close(g[key].cpu(), w[key].cpu(), 2e-5, 'pair %d %s' % (p, key))
