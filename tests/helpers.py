"""Comparison helpers shared by the parity tests."""
import numpy as np
import torch


def sq_dist_f32(q, s):
    """(dx*dx + dy*dy) + dz*dz in float32, the metric both the reference and the kernels use."""
    d = q[:, None, :] - s[None, :, :]
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def assert_neighbors_equal(a, b, q_points, s_points, context=''):
    """Bit-exact equality of two (Nq, W) neighbour tables, except that entries of one row whose float32 squared
    distances are exactly equal may be permuted (the reference orders ties with an unstable std::sort)."""
    a, b = torch.as_tensor(a).long().cpu(), torch.as_tensor(b).long().cpu()
    assert a.shape == b.shape, '%s shape %s vs %s' % (context, tuple(a.shape), tuple(b.shape))
    if torch.equal(a, b):
        return 0
    q, s = torch.as_tensor(q_points).float().cpu(), torch.as_tensor(s_points).float().cpu()
    ns = s.shape[0]
    s_pad = torch.cat((s, torch.full((1, 3), float('inf'))), 0)
    rows = torch.nonzero((a != b).any(1))[:, 0]
    for r in rows.tolist():
        da = ((q[r] - s_pad[a[r]]) ** 2)
        db = ((q[r] - s_pad[b[r]]) ** 2)
        da = (da[:, 0] + da[:, 1]) + da[:, 2]
        db = (db[:, 0] + db[:, 1]) + db[:, 2]
        da[a[r] == ns] = float('inf')
        db[b[r] == ns] = float('inf')
        assert torch.equal(da, db), '%s row %d: distance sequences differ' % (context, r)
        # a permutation inside a tie group can also push a tied entry across the last column, so compare as
        # multisets except for the final tie group
        last = da[-1]
        keep = da != last
        assert sorted(a[r][keep].tolist()) == sorted(b[r][keep].tolist()), '%s row %d: index sets differ' % (context, r)
    return len(rows)


def rel_err(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, rel=1e-4, context=''):
    """max |a-b| <= rel * max |b|  (the 1e-4 relative tolerance BASELINE.json states for fp32 outputs)."""
    e = rel_err(a, b)
    assert e <= rel, '%s: relative error %.3e > %.1e' % (context, e, rel)


def assert_pairs_equal_up_to_ties(idx_a, scores_a, idx_b, scores_b, rtol=1e-5, context=''):
    """Top-k (ref, src) correspondence lists: same pairs in the same order, except among near-equal scores."""
    pa = list(zip(*[torch.as_tensor(x).cpu().tolist() for x in idx_a]))
    pb = list(zip(*[torch.as_tensor(x).cpu().tolist() for x in idx_b]))
    sa, sb = torch.as_tensor(scores_a).double().cpu(), torch.as_tensor(scores_b).double().cpu()
    assert len(pa) == len(pb), context
    np.testing.assert_allclose(sa.numpy(), sb.numpy(), rtol=1e-3, atol=0, err_msg=context)
    i = 0
    n = len(pa)
    boundary = float(sb[-1])
    while i < n:
        j = i + 1
        while j < n and abs(float(sb[j]) - float(sb[i])) <= rtol * abs(float(sb[i])):
            j += 1
        if abs(float(sb[i]) - boundary) > rtol * abs(boundary):       # the last tie group may be cut differently
            assert sorted(pa[i:j]) == sorted(pb[i:j]), '%s: pairs %d..%d differ' % (context, i, j)
        i = j


def index_checksum(t):
    """Order-sensitive checksum of an index array: sum of value * ((flat position mod 65521) + 1) mod 2^61 - 1 -- the formula
    tests/golden/generate_golden.py applied to the reference's tables."""
    v = np.asarray(t).astype(np.uint64).reshape(-1)
    w = (np.arange(v.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return int((v * w).sum() % np.uint64(2 ** 61 - 1))


def table_sq_dists(q, s, t):
    """float32 ((dx*dx + dy*dy) + dz*dz) of every entry of a neighbour table (padding index = len(s): +inf): the metric of
    radius_neighbors_cpu.cpp / nanoflann and of the kernels."""
    q, s, t = np.asarray(q, np.float32), np.asarray(s, np.float32), np.asarray(t).astype(np.int64)
    sp = np.concatenate([s, np.full((1, 3), np.inf, np.float32)], 0)
    d = q[:, None, :] - sp[t]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    d2[t == len(s)] = np.inf
    return d2


def tie_canonical(q, s, t):
    """A neighbour table with the reference's one degree of freedom removed: entries of a row whose float32 squared distances are EQUAL are
    put in ascending index order (the reference leaves them to an unstable std::sort), and the members of the tie group that ends a FULL row
    (it may have been cut by the neighbour limit: which of its members survive is again the sort's choice) are replaced by the bit pattern
    of their distance.  Two tables with the same canonical form differ only inside groups of exactly tied entries.
    -> (canonical int64 table, number of rows holding a tie, number of tied entries)."""
    t = np.asarray(t).astype(np.int64)
    d2 = table_sq_dists(q, s, t)
    if t.shape[1] > 1:
        assert bool((d2[:, 1:] >= d2[:, :-1]).all()), 'a neighbour table must be sorted by distance'
    o1 = np.argsort(t, axis=1, kind='stable')
    t1, d1 = np.take_along_axis(t, o1, 1), np.take_along_axis(d2, o1, 1)
    o2 = np.argsort(d1, axis=1, kind='stable')
    tc, dc = np.take_along_axis(t1, o2, 1), np.take_along_axis(d1, o2, 1)
    tied = np.zeros(t.shape, dtype=bool)
    if t.shape[1] > 1:
        eq = (dc[:, 1:] == dc[:, :-1]) & np.isfinite(dc[:, 1:])
        tied[:, 1:] |= eq
        tied[:, :-1] |= eq
    full = t[:, -1] != len(s)
    last = (dc == dc[:, -1:]) & full[:, None]
    bits = np.ascontiguousarray(dc.astype(np.float32)).view(np.int32).astype(np.int64) + (1 << 40)
    tc = np.where(last, bits, tc)
    return tc, int(tied.any(1).sum()), int(tied.sum())
