"""CPU: se3et_amd.ext (host-pointer variants of geotransformer.ext for DataLoader workers) against the C oracle, the reference's own
build (oracle/_ref, when present) and the reference fixture; callable from forked worker processes."""
import multiprocessing as mp

import numpy as np
import pytest
import torch

from helpers import assert_neighbors_equal


def _pair(preset):
    from se3et_amd.synthetic import make_pair
    ref, src, _ = make_pair(preset)
    return torch.from_numpy(np.concatenate([ref, src], 0)), torch.tensor([len(ref), len(src)])


@pytest.mark.parametrize('preset,voxel', [('micro', 0.025), ('c1_2k', 0.05), ('c2_5k', 0.1)])
def test_host_grid_subsampling_is_bit_exact(preset, voxel):
    from oracle import native
    from se3et_amd import ext
    pts, lens = _pair(preset)
    nrm = torch.randn(pts.shape, generator=torch.Generator().manual_seed(1))
    sp, sl, sn = ext.grid_subsampling(pts, lens, nrm, voxel)
    wp, wl, wn = native.grid_subsample(pts, lens, nrm, voxel)
    assert sl.tolist() == wl.tolist()
    assert torch.equal(sp, wp) and torch.equal(sn, wn)              # points AND the libstdc++ emission order


@pytest.mark.parametrize('preset,radius', [('micro', 0.0625), ('c1_2k', 0.0625), ('c2_5k', 0.125)])
def test_host_radius_neighbors_match_oracle(preset, radius):
    from oracle import native
    from se3et_amd import ext
    pts, lens = _pair(preset)
    got = ext.radius_neighbors(pts, pts, lens, lens, radius)
    want = native.radius_search(pts, pts, lens, lens, radius, got.shape[1] + 5)
    assert bool((want[:, got.shape[1]:] == pts.shape[0]).all())     # the width IS the largest neighbour count
    assert_neighbors_equal(got, want[:, :got.shape[1]], pts, pts, 'host radius search')


def test_host_pyramid_matches_reference_fixture(golden_dir):
    """The reference's collate loop on top of se3et_amd.ext reproduces the fixture pyramid of the C1 pair."""
    from se3et_amd import ext
    d = np.load(golden_dir + '/precompute_c1.npz')
    pts = torch.cat([torch.from_numpy(d['ref']), torch.from_numpy(d['src'])])
    lens = torch.tensor([len(d['ref']), len(d['src'])])
    nrm = torch.zeros_like(pts)
    voxel, radius, limits = 0.025, 0.0625, [38, 36, 36, 38]
    stage_pts, stage_len = [pts], [lens]
    for i in range(1, 4):
        voxel *= 2
        pts, lens, nrm = ext.grid_subsampling(pts, lens, nrm, voxel)
        stage_pts.append(pts)
        stage_len.append(lens)
    for i in range(4):
        assert stage_len[i].tolist() == d['lengths_%d' % i].tolist()
        assert torch.equal(stage_pts[i], torch.from_numpy(d['points_%d' % i]))
        nb = ext.radius_neighbors(stage_pts[i], stage_pts[i], stage_len[i], stage_len[i], radius)[:, :limits[i]]
        assert_neighbors_equal(nb, d['neighbors_%d' % i], stage_pts[i], stage_pts[i], 'neighbors %d' % i)
        radius *= 2


def test_host_ext_rejects_what_the_reference_rejects():
    from se3et_amd import ext
    p, l = torch.zeros(4, 3), torch.tensor([4])
    with pytest.raises(RuntimeError):
        ext.grid_subsampling(p.double(), l, p, 0.1)
    with pytest.raises(RuntimeError):
        ext.radius_neighbors(p, p.t().contiguous().t(), l, l, 0.1)
    with pytest.raises(RuntimeError):
        ext.radius_neighbors(p, p, l.int(), l, 0.1)


def _worker(q):
    from se3et_amd import ext
    pts, lens = _pair('micro')
    q.put(ext.radius_neighbors(pts, pts, lens, lens, 0.0625).numpy())


def test_host_ext_runs_in_forked_worker_processes():
    from se3et_amd import ext
    pts, lens = _pair('micro')
    here = ext.radius_neighbors(pts, pts, lens, lens, 0.0625).numpy()
    ctx = mp.get_context('fork')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(q,)) for _ in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(np.array_equal(o, here) for o in outs)


def _host_collate(ref, src, num_stages, voxel, radius, limits):
    """geotransformer/utils/data.py:13-97 (precompute_data_stack_mode) on top of se3et_amd.ext, as a DataLoader worker would run it."""
    from se3et_amd import ext
    pts = torch.from_numpy(np.concatenate([ref, src], 0))
    lens = torch.tensor([len(ref), len(src)])
    nrm = torch.zeros_like(pts)
    P, L = [pts], [lens]
    for i in range(1, num_stages):
        voxel *= 2
        pts, lens, nrm = ext.grid_subsampling(pts, lens, nrm, voxel)
        P.append(pts)
        L.append(lens)
    out = {'points': P, 'lengths': L, 'neighbors': [], 'subsampling': [], 'upsampling': []}
    for i in range(num_stages):
        out['neighbors'].append(ext.radius_neighbors(P[i], P[i], L[i], L[i], radius)[:, :limits[i]])
        if i < num_stages - 1:
            out['subsampling'].append(ext.radius_neighbors(P[i + 1], P[i], L[i + 1], L[i], radius)[:, :limits[i]])
            out['upsampling'].append(ext.radius_neighbors(P[i], P[i + 1], L[i], L[i + 1], radius * 2)[:, :limits[i + 1]])
        radius *= 2
    return out


def test_host_ext_reproduces_the_reference_tables_of_the_real_pair_including_ties(golden_dir):
    """data/demo (demo_se3ete.npz): 57 % of the stage-0 rows hold exactly tied distances, which the reference orders by an unstable
    std::sort over its k-d tree's visiting order.  se3et_amd.ext restates that tree (csrc/host_ext.hip: KdTree) and sorts with the same
    std::sort: ALL TEN tables of the reference's collate, bit for bit -- the order-sensitive checksums of the fixture -- and the last
    stage's points."""
    from helpers import index_checksum
    g = np.load(golden_dir + '/demo_se3ete.npz')
    dd = _host_collate(g['ref'], g['src'], 4, 0.025, 0.0625, [38, 36, 36, 38])
    assert np.array_equal(np.stack([l.numpy() for l in dd['lengths']]), g['lengths'])
    assert torch.equal(dd['points'][-1], torch.from_numpy(g['points_last']))
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for i, t in enumerate(dd[key]):
            assert t.shape[1] == int(g['width/' + key][i]), '%s[%d] width' % (key, i)
            assert index_checksum(t.numpy()) == int(g['checksum/' + key][i]), '%s[%d]: differs from the reference (order-sensitive checksum)' % (key, i)


def test_host_radius_neighbors_equal_the_reference_build_on_a_lattice():
    """Points on a coarse lattice (almost every distance tied) against the reference's own extension (oracle/_ref, build container only):
    identical tables, whatever the tie order."""
    import os
    from oracle import ref_shims
    if not os.path.exists(ref_shims.REF_EXT_SO):
        pytest.skip('the reference build (oracle/_ref) is not present')
    ref_ext = ref_shims._RefExt(ref_shims.REF_EXT_SO)
    from se3et_amd import ext
    g = torch.Generator().manual_seed(9)
    for n, cell, radius in ((3000, 0.02, 0.0625), (1200, 0.05, 0.11), (40, 0.05, 0.3), (11, 0.01, 0.05), (5, 0.1, 0.25)):
        pts = (torch.randint(0, 24, (2 * n, 3), generator=g).float() * cell).contiguous()
        lens = torch.tensor([n, n])
        got = ext.radius_neighbors(pts, pts, lens, lens, radius)
        want = ref_ext.radius_neighbors(pts, pts, lens, lens, radius)
        assert torch.equal(got, want), (n, cell, radius)
