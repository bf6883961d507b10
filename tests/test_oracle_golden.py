"""CPU: the oracle (oracle/) against fixtures captured from the genuine reference (tests/golden/*.npz)."""
import numpy as np
import pytest
import torch

from helpers import assert_close, assert_neighbors_equal, assert_pairs_equal_up_to_ties


def _cfg(g, **kw):
    from oracle import se3et_oracle as O
    return O.OracleConfig(blocks=[str(b) for b in g['blocks']], **kw)


def _data(g):
    data = {k: [] for k in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling')}
    for key in data:
        i = 0
        while 'data/%s_%d' % (key, i) in g.files:
            t = torch.from_numpy(g['data/%s_%d' % (key, i)])
            data[key].append(t.long() if key != 'points' else t)
            i += 1
    data['features'] = torch.ones((data['points'][0].shape[0], 1))
    return data


def test_tables_match_reference(golden_dir):
    from se3et_amd import tables as T
    d = np.load(golden_dir + '/tables_kanchor6.npz')
    assert np.abs(T.rotations() - d['vRs']).max() < 1e-6
    assert np.array_equal(T.anchors(), d['anchors'])
    to, tr = T.trace_indices()
    assert np.array_equal(to, d['trace_idx_ori']) and np.array_equal(tr, d['trace_idx_rot'])
    assert np.array_equal(T.kernel_points(1.0), d['kernel_points_unit'])
    assert np.array_equal(T.kernel_slot_table(), d['kidx_rot'][:, 0, :])
    assert np.array_equal(T.anchor_slot_table(), d['ridx_rot'][0])
    assert np.abs(T.quotient_anchors() - d['quotient_anchors']).max() < 1e-6
    w0, w1 = T.wigner_tables()
    assert np.array_equal(w0, d['wignerD0']) and np.array_equal(w1, d['wignerD1'])


def test_precompute_c1_matches_reference(golden_dir):
    from oracle import se3et_oracle as O
    d = np.load(golden_dir + '/precompute_c1.npz')
    pts = torch.cat([torch.from_numpy(d['ref']), torch.from_numpy(d['src'])])
    out = O.precompute(pts, torch.tensor([len(d['ref']), len(d['src'])]), 4, 0.025, 0.0625, [38, 36, 36, 38])
    for i in range(4):
        assert out['lengths'][i].tolist() == d['lengths_%d' % i].tolist()
        assert torch.equal(out['points'][i], torch.from_numpy(d['points_%d' % i]))       # incl. unordered_map order
        assert_neighbors_equal(out['neighbors'][i], d['neighbors_%d' % i], out['points'][i], out['points'][i])
    for i in range(3):
        assert_neighbors_equal(out['subsampling'][i], d['subsampling_%d' % i], out['points'][i + 1], out['points'][i])
        assert_neighbors_equal(out['upsampling'][i], d['upsampling_%d' % i], out['points'][i], out['points'][i + 1])


@pytest.mark.parametrize('preset,stages,voxel,radius', [('c2_5k', 4, 0.025, 0.0625), ('c3_20k', 5, 0.3, 1.275)])
def test_precompute_sizes_match_reference(golden_dir, preset, stages, voxel, radius):
    from oracle import se3et_oracle as O
    from se3et_amd.synthetic import make_pair
    d = np.load(golden_dir + '/precompute_sizes.npz')
    ref, src, _ = make_pair(preset)
    pts = torch.from_numpy(np.concatenate([ref, src], 0))
    out = O.precompute(pts, torch.tensor([len(ref), len(src)]), stages, voxel, radius, [38, 36, 36, 38, 38][:stages])
    assert np.array_equal(np.stack([l.numpy() for l in out['lengths']]), d[preset + '/lengths'])
    assert [n.shape[1] for n in out['neighbors']] == d[preset + '/neighbor_widths'].tolist()
    assert [n.shape[1] for n in out['subsampling']] == d[preset + '/subsampling_widths'].tolist()
    assert [n.shape[1] for n in out['upsampling']] == d[preset + '/upsampling_widths'].tolist()
    assert torch.equal(out['points'][-1], torch.from_numpy(d[preset + '/points_last']))


@pytest.mark.parametrize('fixture', ['micro_se3ete.npz', 'micro_se3eti.npz'])
def test_micro_forward_matches_reference(golden_dir, fixture):
    from oracle import se3et_oracle as O
    g = np.load(golden_dir + '/' + fixture)
    state = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    cfg = _cfg(g, init_dim=8, output_dim=32, group_norm=4, gt_hidden_dim=32,
               n_level_equiv=2 if 'transformer.embedding.anchors_wignerD.0' in state else 0)
    with torch.no_grad():
        out = O.forward(state, cfg, _data(g))
    assert_close(out['feats_c'], g['out/feats_c'], 1e-5, 'feats_c')
    assert_close(out['feats_f'], g['out/feats_f'], 1e-5, 'feats_f')
    assert_close(out['ref_feats_c'], g['out/ref_feats_c'], 1e-5, 'ref_feats_c')
    assert_close(out['src_feats_c'], g['out/src_feats_c'], 1e-5, 'src_feats_c')
    ri, si = torch.from_numpy(g['out/ref_node_corr_indices']).long(), torch.from_numpy(g['out/src_node_corr_indices']).long()
    assert_pairs_equal_up_to_ties((out['ref_node_corr_indices'], out['src_node_corr_indices']), out['node_corr_scores'],
                                  (ri, si), out['node_corr_scores'], rtol=1e-5, context=fixture)
    assert_close(out['estimated_transform'], g['out/estimated_transform'], 1e-4, 'estimated_transform')
    assert int(g['out/num_corr']) == out['ref_corr_points'].shape[0]


def test_per_op_fixtures(golden_dir):
    """Single ops of the oracle against inputs/outputs captured with forward hooks inside the reference model."""
    from oracle import se3et_oracle as O
    g = np.load(golden_dir + '/micro_se3ete.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    cfg = _cfg(g, init_dim=8, output_dim=32, group_norm=4, gt_hidden_dim=32)
    with torch.no_grad():
        q, s, idx, x = [torch.from_numpy(g['op/kpconv_2_2/in%d' % i]) for i in range(4)]
        got = O.kpconv_inter_so3(sd, 'backbone.encoder2_2.interso3.conv.', q, s, idx.long(), x, cfg.init_sigma * 2)
        assert_close(got, g['op/kpconv_2_2/out0'], 1e-5, 'kpconv')
        pts = torch.from_numpy(g['op/embedding/in0'])[0]
        emb = O.geometric_embedding(sd, 'transformer.embedding.', pts, cfg)
        assert_close(emb, g['op/embedding/out0'][0], 1e-5, 'geometric embedding')
        eq = O.equiv_embedding(sd, 'transformer.embedding.', pts)
        assert_close(eq, g['op/embedding/out1'][0], 1e-6, 'equivariant embedding')
        x0 = torch.from_numpy(g['op/attn_0/in0'])[0]
        hid, sc = O.rpe_attention(sd, 'transformer.transformer.layers.0.attention.attention.', x0, x0, emb, eq, 4)
        assert_close(hid, g['op/attn_0/out0'][0], 1e-5, 'self_eq hidden')
        assert_close(sc, g['op/attn_0/out1'][0], 1e-5, 'self_eq scores')
        out0, _ = O.rpe_layer(sd, 'transformer.transformer.layers.0.', x0, emb, eq, 4)
        assert_close(out0, g['op/layer_0/out0'][0], 1e-5, 'self_eq layer')
        q1, k1 = torch.from_numpy(g['op/attn_1/in0'])[0], torch.from_numpy(g['op/attn_1/in1'])[0]
        out1, w1 = O.cross_eq_layer(sd, 'transformer.transformer.layers.1.', q1, k1, 4, 'a_soft')
        assert_close(out1, g['op/layer_1/out0'][0], 1e-5, 'cross_a_soft layer')
        q3, k3 = torch.from_numpy(g['op/attn_3/in0'])[0], torch.from_numpy(g['op/attn_3/in1'])[0]
        out3, w3 = O.cross_eq_layer(sd, 'transformer.transformer.layers.3.', q3, k3, 4, 'r_soft')
        assert_close(out3, g['op/layer_3/out0'][0], 1e-5, 'cross_r_soft layer')
        assert_close(w3, g['op/attn_3/out2'].reshape(-1), 1e-5, 'rotation weights')
        sk = O.log_optimal_transport(torch.from_numpy(g['op/sinkhorn/in0']), torch.from_numpy(g['op/sinkhorn/in1']),
                                     torch.from_numpy(g['op/sinkhorn/in2']), sd['optimal_transport.alpha'], 100)
        want = torch.from_numpy(g['op/sinkhorn/out0'])
        valid = want > -1e11
        assert float((sk[valid] - want[valid]).abs().max()) < 1e-5 * float(want[valid].abs().max())


@pytest.mark.parametrize('variant,fixture', [('se3ete2', 'synthw_se3ete2.npz'), ('se3eti2', 'synthw_se3eti2.npz'),
                                             ('se3eti_kitti', 'synthw_se3eti_kitti.npz'),
                                             ('se3eti2', 'synthw_se3eti2_c1.npz')])      # BASELINE.json configs[0]: SE3ET-I2 on the 2k+2k pair
def test_real_width_forward_matches_reference(golden_dir, variant, fixture):
    """Oracle + the product's own state-dict construction (tables, names, shapes) + name-keyed synthetic weights."""
    from oracle import se3et_oracle as O
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    g = np.load(golden_dir + '/' + fixture)
    cfg = make_cfg(variant)
    model = load_synthetic_weights(create_model(cfg), int(g['synth_seed']))
    sd = model.state_dict()
    names = [str(n) for n in g['sd_names']]
    assert set(names) == set(sd.keys())
    for n, shp, dt in zip(names, g['sd_shapes'], g['sd_dtypes']):
        want = tuple(int(v) for v in str(shp).split(',')) if str(shp) else ()
        assert tuple(sd[n].shape) == want and str(sd[n].dtype) == 'torch.' + str(dt), n
    ref, src, _ = make_pair(str(g['pair']))
    pts = torch.from_numpy(np.concatenate([ref, src], 0))
    b = cfg.backbone
    oc = O.OracleConfig.from_model_cfg(cfg)
    data = O.precompute(pts, torch.tensor([len(ref), len(src)]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    assert np.array_equal(np.stack([l.numpy() for l in data['lengths']]), g['lengths'])
    data['features'] = torch.ones((pts.shape[0], 1))
    with torch.no_grad():
        out = O.forward({k: v.detach() for k, v in sd.items()}, oc, data)
    assert_close(out['feats_c'][:, :, :64], g['out/feats_c'], 1e-5, 'feats_c')
    assert_close(out['ref_feats_c'], g['out/ref_feats_c'], 1e-5, 'ref_feats_c')
    assert_close(out['src_feats_c'], g['out/src_feats_c'], 1e-5, 'src_feats_c')
    assert_close(out['estimated_transform'], g['out/estimated_transform'], 1e-3, 'estimated_transform')


def test_precompute_cap_matches_reference(golden_dir):
    """The 2000-superpoint cap of the last stage (geotransformer/utils/data.py:34-43) on the cap_30k pair."""
    from helpers import index_checksum
    from oracle import se3et_oracle as O
    from se3et_amd.synthetic import make_pair
    g = np.load(golden_dir + '/precompute_cap.npz')
    ref, src, _ = make_pair('cap_30k')
    pts = torch.from_numpy(np.concatenate([ref, src], 0))
    out = O.precompute(pts, torch.tensor([len(ref), len(src)]), 4, 0.025, 0.0625, [38, 36, 36, 38])
    assert out['lengths'][-1].tolist() == [2000, 2000]
    assert np.array_equal(np.stack([l.numpy() for l in out['lengths']]), g['lengths'])
    assert torch.equal(out['points'][-1], torch.from_numpy(g['points_last']))
    for key in ('neighbors', 'subsampling', 'upsampling'):
        assert [t.shape[1] for t in out[key]] == g['width/' + key].tolist()
        for i, t in enumerate(out[key]):
            assert index_checksum(np.sort(t.numpy(), 1)) == int(g['rowset/' + key][i]), '%s[%d]' % (key, i)


def test_c2_fullsize_forward_matches_reference(golden_dir):
    """BASELINE.json configs[1] at its own size (SE3ET-E, 5k+5k pair 0, 382 / 304 superpoints): the oracle against the genuine
    reference -- pyramid tables by checksum, every transformer layer, features, correspondences, transform."""
    from helpers import index_checksum
    from oracle import se3et_oracle as O
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    g = np.load(golden_dir + '/c2_se3ete_5k.npz')
    cfg = make_cfg('se3ete')
    sd = {k: v.detach() for k, v in load_synthetic_weights(create_model(cfg), int(g['synth_seed'])).state_dict().items()}
    ref, src, _ = make_pair(str(g['pair']))
    pts = torch.from_numpy(np.concatenate([ref, src], 0))
    b = cfg.backbone
    oc = O.OracleConfig.from_model_cfg(cfg)
    data = O.precompute(pts, torch.tensor([len(ref), len(src)]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    assert np.array_equal(np.stack([l.numpy() for l in data['lengths']]), g['p0/lengths'])
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for i, t in enumerate(data[key]):
            assert index_checksum(t.numpy()) == int(g['checksum/' + key][i]), '%s[%d]' % (key, i)
    data['features'] = torch.ones((pts.shape[0], 1))
    taps = {}
    with torch.no_grad():
        out = O.forward(sd, oc, data, layer_tap=lambda i, t: taps.__setitem__(i, t))
    rs = int(g['row_step'])
    for i, block in enumerate(g['blocks']):
        assert_close(taps[i][..., ::rs, :], g['op/layer_%d/out0' % i][0], 1e-5, 'layer %d (%s)' % (i, block))
    assert_close(out['feats_c'][::rs, :, ::4], g['p0/feats_c'], 1e-5, 'feats_c')
    assert_close(out['ref_feats_c'], g['p0/ref_feats_c'], 1e-5, 'ref_feats_c')
    assert_close(out['src_feats_c'], g['p0/src_feats_c'], 1e-5, 'src_feats_c')
    ri, si = torch.from_numpy(g['p0/ref_node_corr_indices']).long(), torch.from_numpy(g['p0/src_node_corr_indices']).long()
    assert_pairs_equal_up_to_ties((out['ref_node_corr_indices'], out['src_node_corr_indices']), out['node_corr_scores'],
                                  (ri, si), out['node_corr_scores'], rtol=1e-5, context='c2 node correspondences')
    if sorted(zip(ri.tolist(), si.tolist())) == sorted(zip(out['ref_node_corr_indices'].tolist(), out['src_node_corr_indices'].tolist())):
        assert out['ref_corr_points'].shape[0] == int(g['p0/num_corr'])
        assert_close(out['estimated_transform'], g['p0/estimated_transform'], 1e-4, 'estimated_transform')


def test_demo_pair_oracle_matches_reference(golden_dir):
    """The reference's real pair data/demo/{ref,src}.npy (demo_se3ete.npz): the oracle's pyramid -- stage lengths, last-stage points (incl. the
    voxel whose z index is -1 and wraps, grid_subsampling_cpu.cpp:47-49), all ten tables in tie-canonical form (they differ from the
    reference's ONLY inside groups of exactly tied distances: 57 % of the stage-0 rows hold one) -- and, on the reference's choice for the cut
    tie groups and the tied 3-NN rows, every transformer layer, the features and the transform at 1e-5."""
    from helpers import index_checksum, tie_canonical
    from oracle import se3et_oracle as O
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    g = np.load(golden_dir + '/demo_se3ete.npz')
    cfg = make_cfg('se3ete')
    sd = {k: v.detach() for k, v in load_synthetic_weights(create_model(cfg), int(g['synth_seed'])).state_dict().items()}
    ref, src = g['ref'], g['src']
    pts = torch.from_numpy(np.concatenate([ref, src], 0))
    b = cfg.backbone
    data = O.precompute(pts, torch.tensor([len(ref), len(src)]), b.num_stages, b.init_voxel_size, b.init_radius, [38, 36, 36, 38])
    assert np.array_equal(np.stack([l.numpy() for l in data['lengths']]), g['lengths'])
    assert torch.equal(data['points'][-1], torch.from_numpy(g['points_last']))
    P = [p.numpy() for p in data['points']]
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for i, t in enumerate(data[key]):
            q, s = {'neighbors': (P[i], P[i]), 'subsampling': (P[min(i + 1, 3)], P[i]), 'upsampling': (P[i], P[min(i + 1, 3)])}[key]
            canon, rows, _ = tie_canonical(q, s, t.numpy())
            assert index_checksum(canon) == int(g['tiecanon/' + key][i]), '%s[%d]' % (key, i)
            assert rows == int(g['tierows/' + key][i])
            rows = torch.from_numpy(g['patch/%s_%d_rows' % (key, i)]).long()
            if len(rows):
                t[rows] = torch.from_numpy(g['patch/%s_%d_vals' % (key, i)]).long()
            assert index_checksum(np.sort(t.numpy(), 1)) == int(g['rowset/' + key][i]), '%s[%d] row sets on the reference tie choice' % (key, i)
    data['features'] = torch.ones((pts.shape[0], 1))
    taps = {}
    with torch.no_grad():
        out = O.forward(sd, O.OracleConfig.from_model_cfg(cfg), data, layer_tap=lambda i, t: taps.__setitem__(i, t))
    rs = int(g['row_step'])
    for i, block in enumerate(g['blocks']):
        assert_close(taps[i][..., ::rs, :], g['op/layer_%d/out0' % i][0], 1e-5, 'layer %d (%s)' % (i, block))
    assert_close(out['feats_c'][::rs, :, ::4], g['p0/feats_c'], 1e-5, 'feats_c')
    assert_close(out['feats_f'][::4 * rs], g['p0/feats_f'], 1e-5, 'feats_f')
    assert_close(out['ref_feats_c'], g['p0/ref_feats_c'], 1e-5, 'ref_feats_c')
    assert_close(out['src_feats_c'], g['p0/src_feats_c'], 1e-5, 'src_feats_c')
    assert_close(out['estimated_transform'], g['p0/estimated_transform'], 1e-4, 'estimated_transform')
