"""GPU: the reference's ORDER of exactly tied distances in the device radius search (csrc/radius_ties.hip) against
the host twin se3et_amd.ext.radius_neighbors (the reference's k-d tree walk + the real std::sort on the CPU, pinned to the reference's own
build by tests/test_host_ext.py).  Lattice clouds -- coordinates on a millimetre grid, as real scans have them -- where most rows hold ties;
jittered clouds, where none does and the pass must not run."""
import numpy as np
import pytest
import torch

from test_radius_ties_cpu import lattice_cloud

pytestmark = pytest.mark.gpu


def _clouds(n1, n2, step, seed=0):
    s = np.concatenate([lattice_cloud(n1, (0.6, 0.5, 0.3), step, seed + 1), lattice_cloud(n2, (0.5, 0.6, 0.3), step, seed + 2)], 0)
    return torch.from_numpy(s), torch.tensor([n1, n2])


@pytest.mark.parametrize('n1,n2,step,radius,limit', [
    (3000, 2500, 0.001, 0.0625, 38),       # exhaustive kernel, ~40 matches per row, the limit cuts through tie groups
    (9000, 7000, 0.002, 0.0625, 36),       # the uniform-grid kernel (supports above ops.GRID_SEARCH_MIN_SUPPORT)
    (1500, 900, 0.0125, 0.0625, 36),       # a coarse lattice: tie groups of dozens
    (700, 650, 0.02, 0.3, 64),             # limit 64: rows with more than 64 matches are flagged as a whole
    (12, 3, 0.01, 0.05, 5),
])
def test_radius_search_gives_the_reference_order_on_lattice_clouds(n1, n2, step, radius, limit):
    from se3et_amd import ext, ops
    from se3et_amd.modules.ops import radius_search
    s, sl = _clouds(n1, n2, step)
    want = ext.radius_neighbors(s, s, sl, sl, radius)[:, :limit]
    got = radius_search(s.cuda(), s.cuda(), sl, sl, radius, limit)
    assert got.shape == want.shape
    assert torch.equal(got.cpu(), want)                                  # every row, bit for bit, ties included
    # ... and the kernels' own index order does differ here (the test means something)
    saved, ops.RADIUS_REFERENCE_TIES = ops.RADIUS_REFERENCE_TIES, False
    try:
        plain = radius_search(s.cuda(), s.cuda(), sl, sl, radius, limit)
    finally:
        ops.RADIUS_REFERENCE_TIES = saved
    if n1 > 100:
        assert int((plain.cpu() != want).any(1).sum()) > 0


def test_queries_of_another_cloud_and_subsampled_supports():
    """Queries that are not the support (the sub- and up-sampling searches of the pyramid): q from a coarser lattice, and the transpose."""
    from se3et_amd import ext
    from se3et_amd.modules.ops import radius_search
    s, sl = _clouds(4000, 3000, 0.002)
    q = torch.cat([s[:4000][::4], s[4000:][::5]]).contiguous()
    ql = torch.tensor([1000, 600])
    for a, al, b, bl, r, lim in ((q, ql, s, sl, 0.0625, 38), (s, sl, q, ql, 0.125, 36)):
        want = ext.radius_neighbors(a, b, al, bl, r)[:, :lim]
        got = radius_search(a.cuda(), b.cuda(), al, bl, r, lim)
        assert torch.equal(got.cpu(), want)


def test_clouds_without_ties_never_reach_the_tie_pass():
    """Jittered clouds (every BASELINE configuration): the search flags no row, no tree is built, the table is the plain search's."""
    from se3et_amd import ops
    from se3et_amd.synthetic import make_pair
    ref, src, _ = make_pair('c2_5k', index=3)
    pts = torch.from_numpy(np.concatenate([ref, src], 0)).cuda()
    lens = torch.tensor([len(ref), len(src)])
    built = []
    real = ops.ReferenceTree

    class Spy(real):
        def __init__(self, *a, **k):
            built.append(1)
            super().__init__(*a, **k)
    ops.ReferenceTree = Spy
    try:
        full, mc = ops.radius_search_reference_order(pts, pts, lens, lens, 0.0625, 38)
    finally:
        ops.ReferenceTree = real
    plain, _ = ops.radius_neighbors(pts, pts, lens, lens, 0.0625, 38)
    assert not built and torch.equal(full, plain)


def test_pyramid_of_lattice_clouds_equals_the_host_collate():
    """se3et_amd.data.precompute_data_stack_mode on two stacked lattice pairs against the reference's collate loop on the host twin: all ten
    tables of every pair, ties included (the stacked batch marks the columns past a pair's own width with -1)."""
    from se3et_amd import ext
    from se3et_amd.data import precompute_data_stack_mode
    clouds = [lattice_cloud(n, (1.2, 1.0, 0.6), 0.001, 20 + k) for k, n in enumerate((5000, 4200, 3100, 4800))]
    pts = torch.from_numpy(np.concatenate(clouds, 0))
    lens = torch.tensor([len(c) for c in clouds])
    limits = [38, 36, 36, 38]
    dd = precompute_data_stack_mode(pts.cuda(), lens, 4, 0.025, 0.0625, limits)
    # host: pair by pair, the reference's loop (geotransformer/utils/data.py:13-97) on se3et_amd.ext
    offs = [[0], [0], [0], [0]]
    for p in range(2):
        P, Ls = [torch.from_numpy(np.concatenate(clouds[2 * p:2 * p + 2], 0))], [lens[2 * p:2 * p + 2].clone()]
        v = 0.025
        for i in range(1, 4):
            v *= 2
            sp, slen, _ = ext.grid_subsampling(P[-1], Ls[-1], torch.zeros_like(P[-1]), v)
            P.append(sp)
            Ls.append(slen)
        r = 0.0625
        for i in range(4):
            rows = slice(offs[i][-1], offs[i][-1] + P[i].shape[0])
            assert torch.equal(dd['points'][i][rows].cpu(), P[i]), 'stage %d points of pair %d' % (i, p)

            def same(key, stage, table, q_stage, s_stage):
                w = table.shape[1]
                got = dd[key][stage][offs[q_stage][-1]:offs[q_stage][-1] + table.shape[0]].cpu()
                assert bool((got[:, w:] == -1).all()), (key, stage, p)
                got = got[:, :w].clone()
                pad = got == dd['points'][s_stage].shape[0]
                got = got - offs[s_stage][-1]
                got[pad] = P[s_stage].shape[0]
                assert torch.equal(got, table), '%s[%d] of pair %d' % (key, stage, p)
            same('neighbors', i, ext.radius_neighbors(P[i], P[i], Ls[i], Ls[i], r)[:, :limits[i]], i, i)
            if i < 3:
                same('subsampling', i, ext.radius_neighbors(P[i + 1], P[i], Ls[i + 1], Ls[i], r)[:, :limits[i]], i + 1, i)
                same('upsampling', i, ext.radius_neighbors(P[i], P[i + 1], Ls[i], Ls[i + 1], 2 * r)[:, :limits[i + 1]], i, i + 1)
            r *= 2
        for i in range(4):
            offs[i].append(offs[i][-1] + P[i].shape[0])


@pytest.mark.parametrize('n1,n2,step,radius,limit', [(6000, 5000, 0.001, 0.0625, 38), (2500, 2000, 0.01, 0.125, 36)])
def test_radius_search_equals_the_reference_binary(n1, n2, step, radius, limit):
    """The same against the reference's OWN compiled extension where it travelled with the tree (oracle/_ref/libref_ext.so: the reference's
    radius_neighbors_cpu.cpp + nanoflann built by oracle/ref_ext/Makefile in the build container; git-ignored, not gpurun-ignored): the device
    search + tie pass against nanoflann itself, ties included."""
    import os
    from oracle import ref_shims
    if not os.path.exists(ref_shims.REF_EXT_SO):
        pytest.skip('oracle/_ref/libref_ext.so did not travel with this tree')
    from se3et_amd.modules.ops import radius_search
    ext = ref_shims._RefExt(ref_shims.REF_EXT_SO)
    s, sl = _clouds(n1, n2, step, seed=40)
    want = ext.radius_neighbors(s, s, sl, sl, radius)[:, :limit]
    got = radius_search(s.cuda(), s.cuda(), sl, sl, radius, limit)
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize('seed', range(10))
def test_random_degenerate_clouds_equal_the_host_twin(seed):
    """Assorted awkward clouds -- duplicates of every point, all points on a line / in a plane / identical, one cloud empty or a single point,
    queries far outside the support box, limits from 1 to 64 -- through the device search + tie pass against the host twin, every row."""
    from se3et_amd import ext
    from se3et_amd.modules.ops import radius_search
    g = np.random.default_rng(100 + seed)
    step = float(g.choice([0.001, 0.004, 0.01, 0.05]))
    kind = seed % 5
    n = [int(g.integers(1, 2500)), int(g.integers(0 if seed % 3 == 0 else 1, 1800))]
    clouds = []
    for k in n:
        p = g.uniform(0, 1, (k, 3)) * np.array([0.5, 0.4, 0.3])
        if kind == 1:
            p[:, 1:] = 0.2                                   # a line
        elif kind == 2:
            p[:, 2] = 0.1                                    # a plane
        elif kind == 3 and k:
            p[:] = p[0]                                      # one point, k times
        elif kind == 4:
            p = np.repeat(p[:max(k // 2, 1)], 2, axis=0)[:k]  # every point twice
        clouds.append((np.round(p / step) * step).astype(np.float32))
    s = torch.from_numpy(np.concatenate(clouds, 0))
    sl = torch.tensor([len(c) for c in clouds])
    q, ql = s, sl
    if seed % 2:                                             # odd seeds: other queries (every third point per cloud), some far outside the box
        n0 = len(clouds[0])
        parts = [s[:n0][::3], torch.cat([s[n0:][::3], s[n0:][:5] + 10.0])]
        q, ql = torch.cat(parts).contiguous(), torch.tensor([len(parts[0]), len(parts[1])])
    radius = float(g.choice([0.03, 0.0625, 0.15]))
    limit = int(g.choice([1, 5, 36, 38, 64]))
    if s.shape[0] == 0 or q.shape[0] == 0:
        pytest.skip('empty stack')
    want = ext.radius_neighbors(q, s, ql, sl, radius)[:, :limit]
    got = radius_search(q.cuda(), s.cuda(), ql, sl, radius, limit)
    assert got.shape == want.shape, (tuple(got.shape), tuple(want.shape))
    assert torch.equal(got.cpu(), want), 'seed %d kind %d limit %d radius %g step %g sizes %s' % (seed, kind, limit, radius, step, n)


def test_a_row_the_pass_cannot_order_raises():
    """The pass is handed a strip shorter than a flagged row's matches (max_hits below the search's count): the rows it leaves behind are
    counted on the device and the next check raises instead of keeping the index order silently."""
    from se3et_amd import ops
    s, sl = _clouds(3000, 2500, 0.001)
    sc = s.cuda()
    flags = torch.zeros(sc.shape[0] + 1, dtype=torch.int32, device='cuda')
    full, mc = ops.radius_neighbors(sc, sc, sl, sl, 0.0625, 38, ties=(flags[1:], flags[:1]))
    n_tie = int(flags[0])
    assert n_tie > 0
    ops.radius_tie_order(full, sc, sc, sl, sl, 0.0625, flags[1:], n_tie, 3)          # 3 << the ~40 matches of a row
    with pytest.raises(RuntimeError, match='could not be given the reference order'):
        ops.tie_overflow_check()
    ops.tie_overflow_check()                                                            # (the words are consumed: a second check is a no-op)
    ops.radius_tie_order(full, sc, sc, sl, sl, 0.0625, flags[1:], n_tie, int(mc.max()))
    ops.tie_overflow_check()
