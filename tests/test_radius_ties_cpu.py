"""CPU: the code of the device pass that gives rows with EXACTLY tied distances the reference's order (csrc/radius_ties.hip: the walk of the
flattened reference k-d tree + the restatement of libstdc++'s std::sort, one row at a time) run on host memory through
se3_debug_radius_tie_order_host, against the host twin se3_radius_neighbors_host -- the reference's tree walked recursively and sorted by
the real std::sort, itself pinned to the reference's own build and to the order-sensitive checksums of data/demo (tests/test_host_ext.py).
Lattice clouds (coordinates on a millimetre grid, as real scans have them): most rows hold ties, so the row contents depend on every
exchange of the sort."""
import ctypes

import numpy as np
import pytest
import torch


def lattice_cloud(n, extent, step, seed, surface=True):
    g = np.random.default_rng(seed)
    p = g.uniform(0, 1, (n, 3)) * np.asarray(extent)
    if surface:
        p[:, 2] = 0.1 * np.sin(6 * p[:, 0]) * np.cos(5 * p[:, 1]) + 0.2
    return (np.round(p / step) * step).astype(np.float32)


def tie_order_on_host(q, s, ql, sl, radius, limit, rows, max_hits):
    """-> the (Nq, limit) table with `rows` rewritten by the device pass's code on the CPU; other rows hold -7."""
    from se3et_amd._lib import check, lib
    L = lib()
    nq, ns, b = q.shape[0], s.shape[0], len(ql)
    qlen, slen = torch.tensor(ql, dtype=torch.int64), torch.tensor(sl, dtype=torch.int64)
    cap = L.se3_kdtree_max_bytes(ns, b)
    tree = torch.zeros(cap, dtype=torch.uint8)
    used = ctypes.c_size_t(0)
    check(L.se3_kdtree_build_host(s.data_ptr(), ns, slen.data_ptr(), b, tree.data_ptr(), cap, ctypes.byref(used)), 'se3_kdtree_build_host')
    assert 0 < used.value <= cap
    out = torch.full((nq, limit), -7, dtype=torch.int64)
    rows = torch.as_tensor(rows, dtype=torch.int32).contiguous()
    over = ctypes.c_int(0)
    check(L.se3_debug_radius_tie_order_host(q.data_ptr(), nq, s.data_ptr(), ns, qlen.data_ptr(), slen.data_ptr(), b, tree.data_ptr(), float(radius),
                                            limit, rows.data_ptr(), rows.numel(), max_hits, out.data_ptr(), ctypes.byref(over)),
          'se3_debug_radius_tie_order_host')
    return out, over.value


@pytest.mark.parametrize('n1,n2,step,radius,limit', [
    (3000, 2500, 0.001, 0.0625, 38),        # a millimetre lattice, ~40 matches per row: the neighbour limit cuts through tie groups
    (1500, 900, 0.0125, 0.0625, 36),        # a coarse lattice: tie groups of dozens, rows far longer than the sort's threshold of 16
    (700, 650, 0.02, 0.3, 64),              # hundreds of matches per row: partitions, deep recursion
    (12, 3, 0.01, 0.05, 5),                 # clouds smaller than one leaf
])
def test_tie_rows_equal_the_host_twin(n1, n2, step, radius, limit):
    from se3et_amd import ext
    s = torch.from_numpy(np.concatenate([lattice_cloud(n1, (0.6, 0.5, 0.3), step, 1), lattice_cloud(n2, (0.5, 0.6, 0.3), step, 2)], 0))
    q = torch.from_numpy(np.concatenate([lattice_cloud(n1 // 2, (0.6, 0.5, 0.3), step, 3), lattice_cloud(n2 // 3 + 1, (0.5, 0.6, 0.3), step, 4)], 0))
    ql, sl = [n1 // 2, n2 // 3 + 1], [n1, n2]
    want = ext.radius_neighbors(q, s, torch.tensor(ql), torch.tensor(sl), radius)
    width = want.shape[1]
    assert width > 0
    got, over = tie_order_on_host(q, s, ql, sl, radius, limit, np.arange(q.shape[0]), max(width, 1))
    assert over == 0
    k = min(limit, width)
    assert torch.equal(got[:, :k], want[:, :k])                              # every row, ties included, bit for bit
    assert bool((got[:, k:] == s.shape[0]).all())                            # padding = the support size
    # the clouds do hold ties (the test means something): rows whose kept entries contain two equal distances
    d = ((q[:, None, :] - torch.cat((s, torch.full((1, 3), 1e6)))[want[:, :k]]) ** 2)
    d2 = (d[..., 0] + d[..., 1]) + d[..., 2]
    tied_rows = int(((d2[:, 1:] == d2[:, :-1]) & (want[:, 1:k] < s.shape[0])).any(1).sum())
    if n1 > 100:
        assert tied_rows > q.shape[0] // 10, tied_rows


def test_only_listed_rows_are_rewritten_and_overflow_is_reported():
    from se3et_amd import ext
    s = torch.from_numpy(lattice_cloud(800, (0.4, 0.4, 0.2), 0.005, 5))
    want = ext.radius_neighbors(s, s, torch.tensor([800]), torch.tensor([800]), 0.05)
    rows = [3, 77, 799, 400]
    got, over = tie_order_on_host(s, s, [800], [800], 0.05, 20, rows, want.shape[1])
    assert over == 0
    assert torch.equal(got[rows], want[rows, :20])
    rest = np.setdiff1d(np.arange(800), rows)
    assert bool((got[rest] == -7).all())
    # a strip shorter than a row's matches: the row is reported and left alone
    got, over = tie_order_on_host(s, s, [800], [800], 0.05, 20, rows, 3)
    assert over == len(rows) and bool((got == -7).all())


def test_std_sort_restatement_on_adversarial_keys():
    """The sort alone, through rows whose matches ALL tie (a query at the centre of a sphere of lattice points cannot give that; identical
    support points do): every comparison is false, so the result is exactly the sequence of exchanges of libstdc++'s introsort."""
    from se3et_amd import ext
    g = np.random.default_rng(9)
    base = g.uniform(0, 1, (40, 3)).astype(np.float32)
    s = torch.from_numpy(np.repeat(base, 30, axis=0))                         # 40 sites x 30 identical points: all distances tie per site
    q = torch.from_numpy(base)
    want = ext.radius_neighbors(q, s, torch.tensor([40]), torch.tensor([1200]), 0.01)
    got, over = tie_order_on_host(q, s, [40], [1200], 0.01, 30, np.arange(40), want.shape[1])
    assert over == 0 and want.shape[1] >= 30
    assert torch.equal(got, want[:, :30])


@pytest.mark.parametrize('mode', [0, 1])
def test_sort_restatement_equals_libstdcxx(mode):
    """The exchanges of the restated sort (mode 0: std::sort = introsort + final insertion sort; mode 1: its heap-sort fallback =
    std::partial_sort over the whole range) on keys that compare by their high word alone and carry their original position in the low word:
    any difference in the ORDER OF TIED keys shows.  Sizes around the threshold of 16, powers of two, few distinct values, sorted / reversed /
    organ-pipe / median-of-three-killer inputs."""
    from se3et_amd._lib import lib
    L = lib()
    g = np.random.default_rng(17)

    def check(high):
        high = np.asarray(high, dtype=np.uint64)
        keys = np.ascontiguousarray((high << np.uint64(32)) | np.arange(len(high), dtype=np.uint64))
        assert L.se3_debug_std_sort_host(keys.ctypes.data, len(keys), mode) == 0, (mode, len(keys))

    for n in list(range(0, 40)) + [63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 1000, 4096, 5000]:
        for distinct in (1, 2, 3, 7, 50, 10 ** 6):
            check(g.integers(0, distinct, n))
        check(np.arange(n))
        check(np.arange(n)[::-1])
        check(np.minimum(np.arange(n), n - 1 - np.arange(n)))                          # organ pipe
        check(np.arange(n) // 3)
        if n >= 4 and n % 2 == 0:                                                      # Musser's median-of-three killer
            k, a = n // 2, np.zeros(n + 1, dtype=np.int64)
            for i in range(1, k + 1):
                if i % 2 == 1:
                    a[i], a[i + 1] = i, k + i
                a[k + i] = 2 * i
            check(a[1:])
