"""Constant tables of the octahedral anchor group (kanchor = 6, quotient C4, 24 rotations), built from first
principles (no mesh library).  They are the *data* the reference derives at import time with trimesh in
geotransformer/modules/e2pn/vgtk/vgtk/functional/rotation.py:256-381,478-517,566-579 and
geotransformer/modules/e2pn/blocks_epn.py:228-332; values are pinned by tests/golden/tables_kanchor6.npz.

Conventions
  vertices v_a      : +z, +x, +y, -x, -y, -z
  rotations R[4a+j] : Rz(alpha_a) Ry(beta_a) Rz(j*pi/2), which maps e_z to v_a; anchors = R[::4]
  trace_idx_ori[r,a]: index e with R_r v_a = v_e           (24, 6)
  kernel points     : 0.7 * radius * [6 vertices, 8 face normals, centre]     (15, 3)
  kidx[k, r]        : weight slot (orbit under the C4 quotient) of the kernel point that R_r maps onto k
  ridx[a, r]        : anchor b with R_r R_b in R_a * C4
"""
import numpy as np

KANCHOR = 6
NUM_ROTATIONS = 24
NUM_KERNEL_POINTS = 15
NUM_WEIGHT_SLOTS = 6

VERTICES = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=np.float32)
_FACES = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 4, 1], [5, 1, 2], [5, 2, 3], [5, 3, 4], [5, 4, 1]])


def _rz(c, s):
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def _ry(c, s):
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _clean(m):
    """Entries of these matrices are exactly 0 or +-1; remove the 1e-17 residue of cos(pi/2)."""
    return np.round(m).astype(np.float32) + 0.0


def rotations():
    out = np.zeros((KANCHOR, 4, 3, 3), dtype=np.float32)
    for a, v in enumerate(VERTICES.astype(np.float64)):
        cb = v[2]
        sb = np.sqrt(max(0.0, 1.0 - cb * cb))
        if sb > 0:
            ca, sa = v[0] / sb, v[1] / sb
        else:
            ca, sa = (1.0 if cb > 0 else -1.0), 0.0
        for j in range(4):
            g = 0.5 * np.pi * j
            out[a, j] = _clean(_rz(ca, sa) @ _ry(cb, sb) @ _rz(np.cos(g), np.sin(g)))
    return out.reshape(NUM_ROTATIONS, 3, 3)


def anchors():
    return rotations()[::4].copy()


def quotient_anchors():
    return np.stack([_clean(_rz(np.cos(0.5 * np.pi * j), np.sin(0.5 * np.pi * j))) for j in range(4)])


def trace_indices():
    R = rotations().astype(np.float64)
    v = VERTICES.astype(np.float64)
    moved = np.einsum('rij,aj->rai', R, v)
    d = ((moved[:, :, None, :] - v[None, None]) ** 2).sum(-1)        # (r, a, e)
    return d.argmin(2).astype(np.int64), d.argmin(1).astype(np.int64)


def face_normals():
    v = VERTICES.astype(np.float64)
    n = v[_FACES].sum(1)
    return (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)


def kernel_points(radius):
    """(15, 3) float32; the scale is applied in float32 as the reference does (blocks_epn.py:161-171)."""
    pts = np.concatenate([VERTICES, face_normals()], 0)
    pts = pts * np.float32(0.7) * np.float32(radius)
    return np.concatenate([pts, np.zeros((1, 3), np.float32)], 0).astype(np.float32)


def _match(points, targets):
    d = ((points[:, None, :] - targets[None]) ** 2).sum(-1)
    assert d.min(1).max() < 1e-6
    return d.argmin(1)


def kernel_slot_table():
    """kidx (15, 6) int64."""
    kp = kernel_points(1.0).astype(np.float64)
    # orbits of the kernel points under rotations about z by multiples of 90 deg, numbered by first member
    slot = -np.ones(NUM_KERNEL_POINTS, dtype=np.int64)
    nxt = 0
    for k in range(NUM_KERNEL_POINTS):
        if slot[k] < 0:
            for q in quotient_anchors().astype(np.float64):
                slot[_match((q @ kp[k])[None], kp)[0]] = nxt
            nxt += 1
    assert nxt == NUM_WEIGHT_SLOTS
    A = anchors().astype(np.float64)
    kidx = np.zeros((NUM_KERNEL_POINTS, KANCHOR), dtype=np.int64)
    for r in range(KANCHOR):
        src = _match(kp @ A[r], kp)          # rows: R_r^T kp[k]
        kidx[:, r] = slot[src]
    return kidx


def anchor_slot_table():
    """ridx (6, 6) int64: ridx[a, r] = b such that R_r R_b = R_a q for a quotient rotation q."""
    A = anchors().astype(np.float64)
    Q = quotient_anchors().astype(np.float64)
    ridx = np.zeros((KANCHOR, KANCHOR), dtype=np.int64)
    for a in range(KANCHOR):
        coset = np.stack([A[a] @ q for q in Q])
        for r in range(KANCHOR):
            best, best_b = -2.0, -1
            for b in range(KANCHOR):
                prod = A[r] @ A[b]
                c = max(0.5 * (np.trace(m.T @ prod) - 1) for m in coset)
                if c > best + 1e-9:
                    best, best_b = c, b
            assert abs(best - 1) < 1e-6
            ridx[a, r] = best_b
    return ridx


def wigner_tables():
    """[D^0 (6,1,1), D^1 (6,3,3)] at the transposed section anchors; D^1(R) = R in the (x, y, z) basis
    (e3nn convention as restated in DESIGN.md -- unpinned, e3nn is not installable here)."""
    A = anchors()
    return np.ones((KANCHOR, 1, 1), np.float32), np.ascontiguousarray(A.transpose(0, 2, 1))


def collapse_rotation_weights(trace_idx_ori):
    """(R, A) int table -> list of R index arrays usable as M[a, trace[r, a]] += w[r] (r_soft collapse)."""
    return np.asarray(trace_idx_ori, dtype=np.int64)
