"""CPU: the numpy restatements of the EPN toolkit kernels (oracle/vgtk_oracle.py) on hand-checkable cases."""
import numpy as np


def test_ball_query_semantics():
    from oracle import vgtk_oracle as VO
    xyz = np.zeros((1, 3, 6), np.float32)
    xyz[0, 0] = [0.0, 0.1, 5.0, 0.2, 6.0, 0.3]
    q = np.zeros((1, 3, 2), np.float32)
    q[0, 0, 1] = 100.0
    idx = VO.ball_query(q, xyz, 1.0, 4)
    assert idx[0, 0].tolist() == [0, 1, 3, 5]                      # first four in index order
    assert idx[0, 1].tolist() == [0, 0, 0, 0]                      # nothing in range: zeros
    assert VO.ball_query(q, xyz, 0.15, 4)[0, 0].tolist() == [0, 1, 0, 1]          # two hits, repeated cyclically
    assert VO.ball_query(q, xyz, 0.25, 4)[0, 0].tolist() == [0, 1, 3, 0]          # short by exactly one: trailing zero (reference quirk)


def test_furthest_point_sampling_semantics():
    from oracle import vgtk_oracle as VO
    pc = np.zeros((1, 3, 5), np.float32)
    pc[0, 0] = [1.0, 2.0, 10.0, 0.0, 4.0]
    idx = VO.furthest_point_sampling(pc, 4)[0].tolist()
    assert idx == [0, 2, 4, 1]                                       # the origin point (index 3) is never chosen


def test_intra_zpconv_is_a_weighted_anchor_gather():
    from oracle import vgtk_oracle as VO
    feats = np.arange(2 * 1 * 3 * 4, dtype=np.float32).reshape(2, 1, 3, 4)
    nbr = np.array([[0, 1], [2, 3]], np.int32)
    w = np.ones((2, 1, 2), np.float32)
    out = VO.intra_zpconv(nbr, w, feats)
    assert out.shape == (2, 1, 1, 3, 2)
    assert out[0, 0, 0, 0].tolist() == [0 + 1, 2 + 3]


def test_intra_zpconv_restatement_matches_reference_twin(golden_dir):
    """The numpy restatement of intraspherical_conv_forward against the reference's own importable twin
    (vgtk/spconv/functional.py:252-270 intra_zpconv_grouping_naive; fixture from tests/golden/generate_golden.py vgtk)."""
    from oracle import vgtk_oracle as VO
    g = np.load(golden_dir + '/vgtk_ops.npz')
    out = VO.intra_zpconv(g['intra_idx'], g['intra_w'], g['intra_feats'])
    assert out.shape == g['intra_out'].shape
    assert np.abs(out - g['intra_out']).max() <= 1e-5 * np.abs(g['intra_out']).max()
