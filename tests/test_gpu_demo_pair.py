"""GPU parity on the reference's ONLY real data: data/demo/{ref,src,gt}.npy (18 977 + 15 953 points of two indoor scans; caller
experiments/se3ete.3dmatch/demo.py:44-58, neighbour limits [38, 36, 36, 38] at :53) against tests/golden/demo_se3ete.npz -- the genuine
collate and SE3ET-E forward on it (generate_golden.py demo; name-keyed synthetic weights, the released checkpoint is not in the repository).

Real scans differ from the jittered box surfaces of the other fixtures in three ways, all met here:
  * full 38-wide neighbour tables with EXACT float32 distance ties in 57 % of the stage-0 rows (coordinates on a millimetre lattice).  The
    reference orders a tie by an unstable std::sort over the KD-tree's traversal order (nanoflann.hpp:1286-1287) and lets the neighbour limit
    cut through a tie group.  Round 6: the search kernels flag such rows and csrc/radius_ties.hip walks the reference's tree and restates
    its sort for them -- all ten tables equal the reference's BIT FOR BIT (the fixture's order-sensitive checksums), and the forward runs on
    them unpatched at the usual 1e-4.  With ops.RADIUS_REFERENCE_TIES = False the kernels' index order is kept: tie_canonical (helpers.py)
    removes exactly that freedom and the outputs are held to the documented looser bound;
  * a cloud whose own minimum falls one ulp below its voxel origin (src: z index -1): the reference's (size_t)floor(..) wraps, which moves one
    voxel of stage 1 in the unordered_map iteration order (found by this test; csrc/grid_subsample.hip: voxel_index);
  * coordinates of ~3 m, where x^2 - 2xy + y^2 carries ~2e-6 of rounding noise: self distances of up to 1.4 mm instead of 0 on the diagonal
    of the geometric embedding (csrc/common.h: se3_ref_sq_dist restates the expression operation by operation), and exact ties at the cut of
    the 3-nearest-superpoint selection (torch.topk's choice: 3 rows, taken from the fixture for the 1e-4 comparison)."""
import contextlib

import numpy as np
import pytest
import torch

from helpers import assert_close, index_checksum, rel_err, tie_canonical

pytestmark = pytest.mark.gpu
LIMITS = [38, 36, 36, 38]


def _model(seed):
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    cfg = make_cfg('se3ete')
    return cfg, load_synthetic_weights(create_model(cfg), seed).cuda().eval()


def _pyramid(cfg, clouds):
    from se3et_amd.data import precompute_data_stack_mode
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
    b = cfg.backbone
    dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, LIMITS)
    dd['features'] = torch.ones((pts.shape[0], 1), device='cuda')
    return dd


def _geometry(points):
    geo = {}
    for i in range(len(points)):
        geo['neighbors', i] = (points[i], points[i])
    for i in range(len(points) - 1):
        geo['subsampling', i] = (points[i + 1], points[i])
        geo['upsampling', i] = (points[i], points[i + 1])
    return geo


def _patch_reference_tie_choice(dd, g, row_offsets=None, pad_from=None, pad_to=None):
    """Writes the reference's rows into the tables where its unstable sort chose another member of a cut tie group (or, for the upsampling
    tables, another nearest point: column 0 feeds nearest_upsample).  -> number of rows written."""
    n = 0
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for i in range(len(dd[key])):
            rows = torch.from_numpy(g['patch/%s_%d_rows' % (key, i)]).long().cuda()
            if rows.numel():
                dd[key][i][rows] = torch.from_numpy(g['patch/%s_%d_vals' % (key, i)]).long().cuda()
                n += rows.numel()
    return n


@contextlib.contextmanager
def _reference_knn3(g):
    """The 3-nearest-superpoint selection of the demo pair's two clouds replaced by the reference's (fixture patch/knn3_*: they differ in the
    3 rows whose third neighbour is exactly tied, torch.topk's choice) in both entry points of the product: ops.knn3_stack (packed forward)
    and ops.geometric_embedding without a precomputed selection (per-module forward)."""
    from se3et_amd import ops
    real_stack, real_emb = ops.knn3_stack, ops.geometric_embedding
    want = [torch.from_numpy(g['patch/knn3_ref']).long(), torch.from_numpy(g['patch/knn3_src']).long()]
    seen = []

    def stack(points, lengths):
        knn = real_stack(points, lengths)
        o = 0
        for n in [int(v) for v in lengths]:
            for w in want:
                if n == len(w) and torch.equal(points[o:o + n].cpu(), torch.from_numpy(g['points_last'])[:n] if w is want[0] else torch.from_numpy(g['points_last'])[-n:]):
                    differ = (knn[o:o + n].cpu().sort(1)[0] != w.sort(1)[0]).any(1)
                    assert int(differ.sum()) <= 4, 'the 3-NN selections differ in %d rows' % int(differ.sum())
                    knn[o:o + n] = w.to(knn.device)
                    seen.append(n)
            o += n
        return knn

    def emb(points, *args, **kw):          # (points, div_term, w_d, b_d, w_a, b_a, sigma_d, sigma_a, k, wigner_d1, dtype, knn, tables)
        args = list(args)
        if len(args) > 10:
            if args[10] is None:
                args[10] = stack(points.contiguous(), [points.shape[0]])
        elif kw.get('knn') is None:
            kw['knn'] = stack(points.contiguous(), [points.shape[0]])
        return real_emb(points, *args, **kw)
    ops.knn3_stack, ops.geometric_embedding = stack, emb
    try:
        yield
    finally:
        ops.knn3_stack, ops.geometric_embedding = real_stack, real_emb
    assert sorted(set(seen)) == sorted(len(w) for w in want), 'the reference selection was not applied to both clouds: %s' % seen


@contextlib.contextmanager
def _index_ordered_ties():
    from se3et_amd import ops
    saved, ops.RADIUS_REFERENCE_TIES = ops.RADIUS_REFERENCE_TIES, False
    try:
        yield
    finally:
        ops.RADIUS_REFERENCE_TIES = saved


def test_demo_pair_pyramid_equals_the_reference_ties_included(golden_dir):
    """all ten tables of the real pair against the fixture's ORDER-SENSITIVE checksums -- the product path, nothing
    patched.  (19 784 of the 34 930 stage-0 rows hold an exact tie; in index order 13 219 of them come in another order and 115 keep another
    member of a cut tie group.)"""
    g = np.load(golden_dir + '/demo_se3ete.npz')
    cfg, _ = _model(int(g['synth_seed']))
    dd = _pyramid(cfg, [g['ref'], g['src']])
    assert np.array_equal(np.stack([l.numpy() for l in dd['lengths']]), g['lengths'])
    assert torch.equal(dd['points'][-1].cpu(), torch.from_numpy(g['points_last'])), 'stage-3 points (grid subsampling incl. emission order)'
    P = [p.cpu().numpy() for p in dd['points']]
    for (key, i), (q, s) in _geometry(P).items():
        t = dd[key][i].cpu().numpy()
        assert t.shape[1] == int(g['width/' + key][i]), '%s[%d] width' % (key, i)
        assert index_checksum(t) == int(g['checksum/' + key][i]), '%s[%d]: not the reference\'s table (order-sensitive checksum)' % (key, i)
        assert index_checksum(np.sort(t, 1)) == int(g['rowset/' + key][i]), '%s[%d] row sets' % (key, i)


def test_demo_pair_pyramid_in_index_order_differs_only_inside_exact_ties(golden_dir):
    """ops.RADIUS_REFERENCE_TIES = False (the search kernels' own order: ties by index)."""
    g = np.load(golden_dir + '/demo_se3ete.npz')
    cfg, _ = _model(int(g['synth_seed']))
    with _index_ordered_ties():
        dd = _pyramid(cfg, [g['ref'], g['src']])
    assert np.array_equal(np.stack([l.numpy() for l in dd['lengths']]), g['lengths'])
    P = [p.cpu().numpy() for p in dd['points']]
    report = []
    for (key, i), (q, s) in _geometry(P).items():
        t = dd[key][i].cpu().numpy()
        assert t.shape[1] == int(g['width/' + key][i]), '%s[%d] width' % (key, i)
        canon, tie_rows, tie_entries = tie_canonical(q, s, t)
        # same canonical form = the tables agree except for the order / the cut of groups of EXACTLY tied distances
        assert index_checksum(canon) == int(g['tiecanon/' + key][i]), '%s[%d]: differs from the reference outside exact distance ties' % (key, i)
        assert (tie_rows, tie_entries) == (int(g['tierows/' + key][i]), int(g['tieentries/' + key][i]))
        exact = index_checksum(t) == int(g['checksum/' + key][i])
        sets = index_checksum(np.sort(t, 1)) == int(g['rowset/' + key][i])
        # what the fixture recorded about the reference's table against index-ordered ties: rows in another order, rows with another SET
        report.append('%s[%d] %dx%d: %d rows hold ties; bit-exact %s; %d rows in another order, %d rows with another neighbour set' % (
            key, i, t.shape[0], t.shape[1], tie_rows, exact, int(g['orderdiff/%s_%d' % (key, i)]), len(g['patch/%s_%d_rows' % (key, i)])))
        assert exact == (int(g['orderdiff/%s_%d' % (key, i)]) == 0)
        if key != 'upsampling':
            assert sets == (len(g['patch/%s_%d_rows' % (key, i)]) == 0)
    print('\n'.join(report))
    # with the reference's choice written into the set-differing rows the row sets are the reference's
    _patch_reference_tie_choice(dd, g)
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for i, t in enumerate(dd[key]):
            assert index_checksum(np.sort(t.cpu().numpy(), 1)) == int(g['rowset/' + key][i]), '%s[%d] row sets after the patch' % (key, i)


def _check_outputs(out, taps, g, tol, label):
    rs = int(g['row_step'])
    assert_close(out['feats_c'][::rs, :, ::4].cpu(), g['p0/feats_c'], tol, label + ' feats_c')
    assert_close(out['feats_f'][::4 * rs].cpu(), g['p0/feats_f'], tol, label + ' feats_f')
    for i, block in enumerate(g['blocks']):
        if taps is not None:
            assert_close(taps[i][..., ::rs, :].cpu(), g['op/layer_%d/out0' % i], tol, '%s layer %d (%s)' % (label, i, block))
    assert_close(out['ref_feats_c'].cpu(), g['p0/ref_feats_c'], tol, label + ' ref_feats_c')
    assert_close(out['src_feats_c'].cpu(), g['p0/src_feats_c'], tol, label + ' src_feats_c')


def test_demo_pair_forward_matches_reference(golden_dir):
    """Single-pair forward of the product path on the real pair -- NO table is patched (round 6: the pyramid reproduces the reference's tie
    choice on the device): every layer, the features and the transform at the 1e-4 of BASELINE.json.  What is still taken from the fixture:
    the 3 superpoints whose THIRD nearest neighbour is exactly tied -- torch.topk's pick among equal values, which differs between torch's own
    CPU and GPU kernels (the fixture is the reference on the CPU; the reference itself runs that line on the GPU)."""
    g = np.load(golden_dir + '/demo_se3ete.npz')
    cfg, model = _model(int(g['synth_seed']))
    dd = _pyramid(cfg, [g['ref'], g['src']])
    taps = {}
    model.transformer.transformer.layer_tap = lambda i, t: taps.__setitem__(i, t)
    with _reference_knn3(g):
        out = model(dd)
    _check_outputs(out, taps, g, 1e-4, 'demo pair')
    got = set(zip(out['ref_node_corr_indices'].tolist(), out['src_node_corr_indices'].tolist()))
    want = set(zip(g['p0/ref_node_corr_indices'].tolist(), g['p0/src_node_corr_indices'].tolist()))
    assert len(got & want) >= 254, 'superpoint correspondences: %d of 256 in common' % len(got & want)
    n_got, n_want = out['ref_corr_points'].shape[0], int(g['p0/num_corr'])
    if got == want and n_got == n_want:
        assert_close(out['estimated_transform'].cpu(), g['p0/estimated_transform'], 1e-4, 'estimated_transform')
    else:          # a correspondence flipped at a threshold: the transform is fitted to a set that differs by it (test_gpu_fullsize.py)
        assert_close(out['estimated_transform'].cpu(), g['p0/estimated_transform'], 5e-3,
                     'estimated_transform (%d of 256 superpoint pairs in common, %d vs %d correspondences)' % (len(got & want), n_got, n_want))


def test_demo_pair_with_index_ordered_ties_stays_close_to_the_reference(golden_dir):
    """ops.RADIUS_REFERENCE_TIES = False (ties in ascending index order, the product path of rounds 1-5): 115 + 6 rows of the neighbour tables, 25 + 1 of the subsampling and
    17 + 1 of the upsampling tables hold another member of a cut tie group than the reference's unstable sort kept, 178 fine points are
    upsampled from another equidistant superpoint, 3 superpoints have another equidistant third neighbour.  These are different, equally
    valid, neighbourhoods: the features move by what one neighbour in 38 is worth.  Measured: backbone 6e-3, transformer layers 2.7e-2 ->
    1.5e-3, final features 1.4e-3 (max-norm relative); held to 5e-2 / 1e-2."""
    g = np.load(golden_dir + '/demo_se3ete.npz')
    cfg, model = _model(int(g['synth_seed']))
    with _index_ordered_ties():
        dd = _pyramid(cfg, [g['ref'], g['src']])
    out = model(dd)
    rs = int(g['row_step'])
    assert_close(out['feats_c'][::rs, :, ::4].cpu(), g['p0/feats_c'], 2e-2, 'feats_c')
    assert_close(out['feats_f'][::4 * rs].cpu(), g['p0/feats_f'], 2e-2, 'feats_f')
    assert_close(out['ref_feats_c'].cpu(), g['p0/ref_feats_c'], 1e-2, 'ref_feats_c')
    assert_close(out['src_feats_c'].cpu(), g['p0/src_feats_c'], 1e-2, 'src_feats_c')
    got = set(zip(out['ref_node_corr_indices'].tolist(), out['src_node_corr_indices'].tolist()))
    want = set(zip(g['p0/ref_node_corr_indices'].tolist(), g['p0/src_node_corr_indices'].tolist()))
    assert len(got & want) >= 240, 'superpoint correspondences: %d of 256 in common' % len(got & want)


def test_demo_pair_as_pair_0_of_a_mixed_batch(golden_dir):
    """The demo pair stacked with seven synthetic 5k+5k pairs through ONE forward (se3et_amd.batched.forward_pairs): its tables must be the
    reference's single-pair tables (order-sensitive checksums, ties included), and its outputs the reference's at 1e-4; the synthetic pairs
    keep their own reference outputs (c2_se3ete_5k.npz) beside a pair six times their size."""
    from se3et_amd.batched import forward_pairs
    from se3et_amd.synthetic import make_pair
    g = np.load(golden_dir + '/demo_se3ete.npz')
    g2 = np.load(golden_dir + '/c2_se3ete_5k.npz')
    cfg, model = _model(int(g['synth_seed']))
    clouds = [g['ref'], g['src']]
    for p in range(1, 8):
        ref, src, _ = make_pair('c2_5k', index=p)
        clouds += [ref, src]
    dd = _pyramid(cfg, clouds)
    n_stage = [int(g['lengths'][s].sum()) for s in range(4)]
    # the demo pair's rows of the stacked tables ARE the reference's single-pair tables, ties included (indices of pair 0 are unshifted;
    # padding = the stacked support size instead of the pair's; columns beyond the pair's own width are marked -1); nothing is patched
    P = [p.cpu().numpy() for p in dd['points']]
    for (key, i), _ in _geometry(P).items():
        qs, ss = {'neighbors': (i, i), 'subsampling': (i + 1, i), 'upsampling': (i, i + 1)}[key]
        t = dd[key][i][:n_stage[qs]].cpu().numpy()
        w = int(g['width/' + key][i])
        assert (t[:, w:] == -1).all(), '%s[%d]: columns beyond the pair\'s own width' % (key, i)
        t = t[:, :w].copy()
        t[t == P[ss].shape[0]] = n_stage[ss]
        assert index_checksum(t) == int(g['checksum/' + key][i]), '%s[%d] of the demo pair inside the batch (order-sensitive)' % (key, i)
    with _reference_knn3(g):
        outs = forward_pairs(model, dd)
    _check_outputs(outs[0], None, g, 1e-4, 'demo pair in a batch of 8')
    for p in (1, 7):
        assert_close(outs[p]['ref_feats_c'][::int(g2['row_step'])].cpu(), g2['p%d/ref_feats_c' % p], 1e-4, 'synthetic pair %d beside the demo pair' % p)
        assert_close(outs[p]['feats_f'][::16 * int(g2['row_step'])].cpu(), g2['p%d/feats_f' % p], 1e-4, 'synthetic pair %d feats_f' % p)
