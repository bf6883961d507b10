"""GPU parity: HIP grid-subsample / radius-search (through the C ABI) vs the C oracle and the reference fixtures."""
import numpy as np
import pytest
import torch

from helpers import assert_neighbors_equal

pytestmark = pytest.mark.gpu


def _clouds(n1, n2, scale, seed):
    g = np.random.default_rng(seed)
    pts = (g.uniform(0, 1, (n1 + n2, 3)) * scale).astype(np.float32)
    nrm = g.normal(size=(n1 + n2, 3)).astype(np.float32)
    return torch.from_numpy(pts), torch.from_numpy(nrm), torch.tensor([n1, n2])


@pytest.mark.parametrize('n1,n2,scale,voxel', [(5000, 4000, 1.0, 0.05), (20000, 15000, 2.0, 0.05),
                                               (3000, 100, 0.5, 0.1), (40000, 40000, 30.0, 0.6), (7, 1, 0.2, 0.05),
                                               (1000, 0, 1.0, 0.1)])
def test_grid_subsample_matches_oracle(n1, n2, scale, voxel):
    from oracle import native
    from se3et_amd import ops
    pts, nrm, lens = _clouds(n1, n2, scale, 0)
    sp, sl, sn = native.grid_subsample(pts, lens, nrm, voxel)
    gp, gn, gl = ops.grid_subsample(pts.cuda(), lens, nrm.cuda(), voxel)
    gl = gl.cpu()
    assert gl.tolist() == sl.tolist()
    m = int(gl.sum())
    assert torch.equal(gp[:m].cpu(), sp), 'points differ (selection or emission order)'
    assert torch.equal(gn[:m].cpu(), sn)


@pytest.mark.parametrize('n1,n2,scale,voxel', [(5000, 4000, 1.0, 0.05), (20000, 15000, 2.0, 0.05), (3000, 100, 0.5, 0.1), (7, 1, 0.2, 0.05),
                                               (1000, 0, 1.0, 0.1)])
def test_grid_subsample_chain_on_device_lengths_matches_oracle(n1, n2, scale, voxel):
    """Three stages back to back, every stage reading the counts of the one before from device memory (se3_grid_subsample_dev: no host
    synchronisation between the stages, the outputs sized by the first stage's rows), against the C oracle run stage by stage: bit-equal
    points, normals and counts -- also with spare rows behind the clouds of the first stage."""
    from oracle import native
    from se3et_amd import ops
    pts, nrm, lens = _clouds(n1, n2, scale, 1)
    want, p, n_, l = [], pts, nrm, lens
    for k in range(3):
        p, l, n_ = native.grid_subsample(p, l, n_, voxel * 2 ** k)
        want.append((p, l, n_))
    spare = torch.full((37, 3), 1e30)
    gp, gn, gl = torch.cat((pts, spare)).cuda(), torch.cat((nrm, spare)).cuda(), lens.cuda()          # device lengths from the start
    got = []
    for k in range(3):
        gp, gn, gl = ops.grid_subsample(gp, gl, gn, voxel * 2 ** k)
        assert gp.shape[0] == n1 + n2 + 37
        got.append((gp, gn, gl))
    for k, ((sp, sl, sn), (gp, gn, gl)) in enumerate(zip(want, got)):
        assert gl.cpu().tolist() == sl.tolist(), 'stage %d counts' % k
        m = int(sl.sum())
        assert torch.equal(gp[:m].cpu(), sp), 'stage %d points' % k
        assert torch.equal(gn[:m].cpu(), sn), 'stage %d normals' % k


@pytest.mark.parametrize('n1,n2,scale,radius,limit', [(3000, 2500, 1.0, 0.08, 38), (6000, 5000, 1.0, 0.0625, 36),
                                                      (500, 3, 0.3, 0.2, 64), (2000, 2000, 0.2, 0.1, 38)])
def test_radius_search_matches_oracle(n1, n2, scale, radius, limit):
    from oracle import native
    from se3et_amd import ops
    pts, _, lens = _clouds(n1, n2, scale, 1)
    g = np.random.default_rng(5)
    q = torch.from_numpy((g.uniform(0, 1, (n1 // 2 + n2 // 2, 3)) * scale).astype(np.float32))
    qlens = torch.tensor([n1 // 2, n2 // 2])
    want = native.radius_search(q, pts, qlens, lens, radius, limit)
    got, mc = ops.radius_neighbors(q.cuda(), pts.cuda(), qlens, lens, radius, limit)
    width = min(limit, int(mc.max()))          # mc: per-cloud maxima
    assert width == want.shape[1]
    assert_neighbors_equal(got[:, :width].cpu(), want, q, pts, 'radius_search')


@pytest.mark.parametrize('ns,nq,radius', [(6000, 3000, 0.08), (20000, 7000, 0.05), (1600, 50, 0.3)])
def test_grid_and_exhaustive_search_agree(ns, nq, radius):
    """The uniform-grid kernel against the exhaustive kernel, with queries partly OUTSIDE the support bounding box."""
    from se3et_amd import ops
    g = np.random.default_rng(7)
    s = torch.from_numpy(g.uniform(0, 1, (ns, 3)).astype(np.float32)).cuda()
    q = torch.from_numpy(g.uniform(-0.2, 1.2, (nq, 3)).astype(np.float32)).cuda()
    sl, ql = torch.tensor([ns // 2, ns - ns // 2]), torch.tensor([nq // 3, nq - nq // 3])
    grid = ops.RadiusGrid(s, sl, radius)
    a, ca = grid.search(q, ql, 40)
    old, ops.GRID_SEARCH_MIN_SUPPORT = ops.GRID_SEARCH_MIN_SUPPORT, 10 ** 12
    try:
        b, cb = ops.radius_neighbors(q, s, ql, sl, radius, 40)
    finally:
        ops.GRID_SEARCH_MIN_SUPPORT = old
    assert ca.tolist() == cb.tolist()
    assert torch.equal(a, b)


def test_precompute_matches_reference_fixture(golden_dir):
    """Whole stage pyramid of the C1 pair against arrays captured from the genuine reference collate."""
    from se3et_amd.modules.ops import grid_subsample, radius_search
    d = np.load(golden_dir + '/precompute_c1.npz')
    pts = torch.cat([torch.from_numpy(d['ref']), torch.from_numpy(d['src'])]).cuda()
    lengths = torch.tensor([len(d['ref']), len(d['src'])])
    voxel, radius = 0.025, 0.0625
    limits = [38, 36, 36, 38]
    points_list, lengths_list = [pts], [lengths]
    for i in range(1, 4):
        voxel *= 2
        p, l, _ = grid_subsample(points_list[-1], lengths_list[-1], torch.zeros_like(points_list[-1]), voxel)
        points_list.append(p)
        lengths_list.append(l)
    for i in range(4):
        assert lengths_list[i].tolist() == d['lengths_%d' % i].tolist()
        assert torch.equal(points_list[i].cpu(), torch.from_numpy(d['points_%d' % i]))
    for i in range(4):
        nb = radius_search(points_list[i], points_list[i], lengths_list[i], lengths_list[i], radius, limits[i])
        assert_neighbors_equal(nb, d['neighbors_%d' % i], points_list[i], points_list[i], 'neighbors_%d' % i)
        if i < 3:
            sub = radius_search(points_list[i + 1], points_list[i], lengths_list[i + 1], lengths_list[i], radius, limits[i])
            assert_neighbors_equal(sub, d['subsampling_%d' % i], points_list[i + 1], points_list[i], 'subsampling_%d' % i)
            up = radius_search(points_list[i], points_list[i + 1], lengths_list[i], lengths_list[i + 1], radius * 2,
                               limits[i + 1])
            assert_neighbors_equal(up, d['upsampling_%d' % i], points_list[i], points_list[i + 1], 'upsampling_%d' % i)
        radius *= 2
