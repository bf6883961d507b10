"""GPU: the EPN toolkit equivalents (se3et_amd/vgtk.py -> csrc/vgtk_ops.hip) against the reference's importable PyTorch
twins (tests/golden/vgtk_ops.npz) and against the numpy restatements of the CUDA sources (oracle/vgtk_oracle.py); determinism of the
sums that the reference computes with atomicAdd."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def test_inter_zpconv_grouping_matches_reference_twin(golden_dir):
    from se3et_amd import vgtk
    g = np.load(golden_dir + '/vgtk_ops.npz')
    idx, w = torch.from_numpy(g['inter_idx']).cuda(), torch.from_numpy(g['inter_w']).cuda()
    b, p, nn_ = idx.shape
    a, ks = w.shape[2], w.shape[3]
    full = idx[:, :, None, None, :].expand(b, p, a, ks, nn_).contiguous()             # the CUDA op takes one index per (p, a, k, n)
    feats = torch.from_numpy(g['feats']).cuda().requires_grad_(True)
    out = vgtk.inter_zpconv_grouping(full, w, feats)
    assert_close(out.detach().cpu(), g['inter_out'], 1e-5, 'inter zpconv forward')
    (out * torch.from_numpy(g['cotangent']).cuda()).sum().backward()
    assert_close(feats.grad.cpu(), g['feats_grad'], 1e-5, 'inter zpconv backward')
    # bit-identical on repetition (the reference's atomicAdd scatter is not)
    f2 = torch.from_numpy(g['feats']).cuda().requires_grad_(True)
    out2 = vgtk.inter_zpconv_grouping(full, w, f2)
    (out2 * torch.from_numpy(g['cotangent']).cuda()).sum().backward()
    assert torch.equal(out2, out) and torch.equal(f2.grad, feats.grad)


def test_gather_points_matches_reference_twin(golden_dir):
    from se3et_amd import vgtk
    g = np.load(golden_dir + '/vgtk_ops.npz')
    pts = torch.from_numpy(g['points']).cuda().requires_grad_(True)
    idx = torch.from_numpy(g['gather_idx']).cuda()
    out = vgtk.Gathering.apply(pts, idx)
    assert torch.equal(out.detach().cpu(), torch.from_numpy(g['gathered']))
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).cuda()
    (out * cot).sum().backward()
    ref = torch.from_numpy(g['points']).cuda().requires_grad_(True)
    (torch.gather(ref, 2, idx.long()[:, None, :].expand(-1, ref.shape[1], -1)) * cot).sum().backward()
    assert_close(pts.grad.cpu(), ref.grad.cpu(), 1e-6, 'gather backward')


@pytest.mark.parametrize('n,m,nsample,radius', [(200, 50, 16, 0.35), (1000, 300, 32, 0.2), (70, 70, 8, 0.05), (33, 5, 4, 10.0)])
def test_ball_query_matches_source_restatement(n, m, nsample, radius):
    from oracle import vgtk_oracle as VO
    from se3et_amd import vgtk
    g = np.random.default_rng(3)
    xyz = g.uniform(0, 1, (2, 3, n)).astype(np.float32)
    new = np.ascontiguousarray(xyz[:, :, :m]) + g.normal(0, 0.01, (2, 3, m)).astype(np.float32)
    got = vgtk.ball_query_index(torch.from_numpy(new).cuda(), torch.from_numpy(xyz).cuda(), radius, nsample).cpu().numpy()
    want = VO.ball_query(new, xyz, radius, nsample)
    # entries whose squared distance is within float round-off of r^2 may fall on either side (the CUDA build contracts to FMAs)
    same = (got == want).all(-1)
    assert same.mean() > 0.98
    for bi, j in zip(*np.nonzero(~same)):
        d2 = ((new[bi, :, j:j + 1] - xyz[bi]) ** 2).sum(0)
        assert np.abs(d2 - radius * radius).min() < 1e-5


@pytest.mark.parametrize('n,m', [(500, 64), (1024, 200), (37, 10), (2500, 128)])
def test_furthest_point_sampling_matches_source_restatement(n, m):
    from oracle import vgtk_oracle as VO
    from se3et_amd import vgtk
    g = np.random.default_rng(4)
    pc = g.normal(0, 1, (2, 3, n)).astype(np.float32)
    pc[1, :, 5] = 0.0                                     # a point at the origin is never selected (|p|^2 <= 1e-3)
    got = vgtk.furthest_sample_index(torch.from_numpy(pc).cuda(), m).cpu().numpy()
    want = VO.furthest_point_sampling(pc, m)
    assert np.array_equal(got, want)
    assert 5 not in got[1].tolist()
    assert len(set(got[0].tolist())) == m


def test_furthest_point_sampling_tie_order():
    """Exact distance ties (points on an integer lattice) resolve as in the reference's block reduction."""
    from oracle import vgtk_oracle as VO
    from se3et_amd import vgtk
    ax = np.arange(6, dtype=np.float32) + 1.0
    pc = np.stack(np.meshgrid(ax, ax, ax, indexing='ij'), 0).reshape(1, 3, -1)          # 216 lattice points
    got = vgtk.furthest_sample_index(torch.from_numpy(pc).cuda(), 40).cpu().numpy()
    assert np.array_equal(got, VO.furthest_point_sampling(pc, 40))


def test_intra_zpconv_grouping_matches_reference_twin(golden_dir):
    """Pinned: the reference's importable twin vgtk/spconv/functional.py:252-270 intra_zpconv_grouping_naive (fixture: forward + gradient),
    1e-5 both ways; bit-identical on repetition (the reference's kernel adds with atomicAdd)."""
    from se3et_amd import vgtk
    g = np.load(golden_dir + '/vgtk_ops.npz')
    nbr, w = torch.from_numpy(g['intra_idx']).cuda(), torch.from_numpy(g['intra_w']).cuda()
    feats = torch.from_numpy(g['intra_feats']).cuda().requires_grad_(True)
    out = vgtk.intra_zpconv_grouping(nbr, w, feats)
    assert_close(out.detach().cpu(), g['intra_out'], 1e-5, 'intra zpconv forward vs reference twin')
    (out * torch.from_numpy(g['intra_cotangent']).cuda()).sum().backward()
    assert_close(feats.grad.cpu(), g['intra_feats_grad'], 1e-5, 'intra zpconv backward vs reference twin')
    f2 = torch.from_numpy(g['intra_feats']).cuda().requires_grad_(True)
    out2 = vgtk.intra_zpconv_grouping(nbr, w, f2)
    (out2 * torch.from_numpy(g['intra_cotangent']).cuda()).sum().backward()
    assert torch.equal(out2, out) and torch.equal(f2.grad, feats.grad)


def test_intra_zpconv_grouping_matches_source_restatement():
    from oracle import vgtk_oracle as VO
    from se3et_amd import vgtk
    g = np.random.default_rng(6)
    b, c, p, na_in, na_out, ks, ann = 2, 5, 31, 12, 12, 3, 4
    nbr = g.integers(0, na_in, (na_out, ann)).astype(np.int32)
    w = g.uniform(0, 1, (na_out, ks, ann)).astype(np.float32)
    feats = torch.from_numpy(g.normal(0, 1, (b, c, p, na_in)).astype(np.float32)).cuda().requires_grad_(True)
    out = vgtk.intra_zpconv_grouping(torch.from_numpy(nbr).cuda(), torch.from_numpy(w).cuda(), feats)
    assert_close(out.detach().cpu(), VO.intra_zpconv(nbr, w, feats.detach().cpu().numpy()), 1e-5, 'intra zpconv forward')
    cot = torch.from_numpy(g.normal(0, 1, tuple(out.shape)).astype(np.float32)).cuda()
    (out * cot).sum().backward()
    ref = feats.detach().clone().requires_grad_(True)
    (torch.einsum('bcpan,akn->bckpa', ref[:, :, :, torch.from_numpy(nbr).long().cuda()], torch.from_numpy(w).cuda()) * cot).sum().backward()
    assert_close(feats.grad.cpu(), ref.grad.cpu(), 1e-5, 'intra zpconv backward')


def test_anchor_queries_match_the_source_restatements():
    """grouping.anchor_query / initial_anchor_query (no PyTorch twin in the reference: restated from the .cu sources in oracle/vgtk_oracle.py)."""
    from oracle import vgtk_oracle as VO
    from se3et_amd import vgtk
    rng = np.random.default_rng(3)
    g = (rng.standard_normal((2, 3, 37, 9)) * 0.3).astype(np.float32)
    anchors = rng.standard_normal((12, 3)).astype(np.float32)
    anchors /= np.linalg.norm(anchors, axis=1, keepdims=True)
    kp = np.stack([rng.uniform(0.1, 0.6, 5), rng.uniform(0.0, 1.2, 5)], 1).astype(np.float32)
    got, = vgtk.anchor_query(None, None, torch.from_numpy(g).cuda(), torch.from_numpy(anchors).cuda(), torch.from_numpy(kp).cuda(), 0)
    want = VO.anchor_query(g, anchors, kp)
    ok = np.isfinite(want)
    assert ok.mean() > 0.99
    assert_close(torch.from_numpy(np.where(ok, got.cpu().numpy(), 0)), np.where(ok, want, 0), 1e-4, 'anchor_query')
    centers = (rng.uniform(-0.5, 0.5, (2, 3, 11))).astype(np.float32)
    xyz = rng.uniform(-0.6, 0.6, (300, 3)).astype(np.float32)
    kpts = (rng.standard_normal((4, 6, 3)) * 0.1).astype(np.float32)
    w, n = vgtk.initial_anchor_query(torch.from_numpy(xyz).cuda(), torch.from_numpy(centers).cuda(), torch.from_numpy(kpts).cuda(), 0.35, 0.06)
    ww, wn = VO.initial_anchor_query(centers, xyz, kpts, 0.35, 0.06)
    assert float(wn.sum()) > 100
    assert torch.equal(n.cpu(), torch.from_numpy(wn))
    assert_close(w.cpu(), ww, 1e-5, 'initial_anchor_query weights')
    w2, n2 = vgtk.initial_anchor_query(torch.from_numpy(xyz).cuda(), torch.from_numpy(centers).cuda(), torch.from_numpy(kpts).cuda(), 0.35, 0.06)
    assert torch.equal(w2, w) and torch.equal(n2, n)          # deterministic (the reference adds with atomicAdd)


def test_gather_points_backward_is_parallel_and_deterministic():
    """Many sources per target, several channel groups and index chunks: the gradient equals index_add in float64 and repeats bit for bit."""
    from se3et_amd import vgtk
    g = torch.Generator().manual_seed(2)
    b, c, n, m = 3, 37, 500, 5000
    pts = torch.randn(b, c, n, generator=g).cuda().requires_grad_(True)
    idx = torch.randint(0, n, (b, m), generator=g, dtype=torch.int32).cuda()
    cot = torch.randn(b, c, m, generator=g).cuda()
    (vgtk.Gathering.apply(pts, idx) * cot).sum().backward()
    want = torch.zeros(b, c, n, dtype=torch.float64, device='cuda')
    for bi in range(b):
        want[bi].index_add_(1, idx[bi].long(), cot[bi].double())
    assert_close(pts.grad.cpu(), want.float().cpu(), 1e-5, 'gather_points backward')
    first = pts.grad.clone()
    pts.grad = None
    (vgtk.Gathering.apply(pts, idx) * cot).sum().backward()
    assert torch.equal(pts.grad, first)
