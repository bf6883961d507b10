"""MI355X equivalents of the EPN toolkit's CUDA-backed functions.

Mirror of the call sites of vgtk.cuda.{gathering, grouping, zpconv} in the reference:
    vgtk/spconv/functional.py:102-129  Gathering (autograd)            -> Gathering, gather_points
    vgtk/pc/sample.py:54-60            ball_query_index                -> ball_query_index
    vgtk/pc/sample.py:62-80            furthest_sample_index           -> furthest_sample_index
    vgtk/so3conv/functional.py:108-109 initial_anchor_query            -> initial_anchor_query
    vgtk/spconv/functional.py:512      cuda_nn.anchor_query            -> anchor_query
    vgtk/spconv/functional.py:314-335  InterZPConvGrouping (autograd)  -> InterZPConvGrouping, inter_zpconv_grouping
    vgtk/spconv/functional.py:211-238  IntraZPConvGrouping (autograd)  -> IntraZPConvGrouping, intra_zpconv_grouping
Same tensor layouts (channel-first float32, int32 indices), CUDA tensors only (on ROCm 'cuda' is the HIP device).  No SE3ET model
calls these; they exist because BASELINE.json lists the vgtk CUDA ops as a replaced subsystem.

Host-side constants (no kernel), for the reference's experiments/<variant>/loss.py:88-145, which builds its rotation-matching loss from the
toolkit at construction time even though every SE3ET config leaves it switched off (se3et_amd.dropin aliases this module as `vgtk`,
`vgtk.functional`, `vgtk.so3conv`):
    vgtk/so3conv/functional.py:398-399   get_octahedron_vertices        -> get_octahedron_vertices   (se3et_amd.tables)
    vgtk/functional/rotation.py:566-579  get_relativeV_index            -> get_relativeV_index
    vgtk/functional/rotation.py:922-934  label_relative_rotation_simple -> label_relative_rotation_simple"""
import numpy as np
import torch

from . import ops as _ops
from ._lib import check, lib

_req, _stream = _ops._req, _ops._stream


def gather_points(points, idx):
    """points (b, c, n) float32, idx (b, m) int32 -> (b, c, m)."""
    points, idx = _req(points.contiguous(), torch.float32, 'points', 3), _req(idx.contiguous(), torch.int32, 'idx', 2)
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    check(lib().se3_vgtk_gather_points_fwd(points.data_ptr(), idx.data_ptr(), b, c, n, m, out.data_ptr(), _stream()), 'se3_vgtk_gather_points_fwd')
    return out


class Gathering(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        ctx.save_for_backward(idx)
        ctx.n = points.shape[2]
        return gather_points(points, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        g = _req(grad_out.contiguous(), torch.float32, 'grad_out', 3)
        b, c, m = g.shape
        grad = torch.empty((b, c, ctx.n), dtype=torch.float32, device=g.device)
        check(lib().se3_vgtk_gather_points_bwd(g.data_ptr(), idx.data_ptr(), b, c, ctx.n, m, grad.data_ptr(), _stream()), 'se3_vgtk_gather_points_bwd')
        return grad, None


def ball_query_index(query_points, support_points, radius, n_sample):
    """query (b, 3, m), support (b, 3, n) -> idx (b, m, n_sample) int32: the first n_sample support indices within `radius`."""
    q, s = _req(query_points.contiguous(), torch.float32, 'query_points', 3), _req(support_points.contiguous(), torch.float32, 'support_points', 3)
    b, _, m = q.shape
    n = s.shape[2]
    if q.shape[1] != 3 or s.shape[1] != 3 or s.shape[0] != b:
        raise RuntimeError('ball_query_index: points must be (b, 3, n)')
    idx = torch.empty((b, m, int(n_sample)), dtype=torch.int32, device=q.device)
    check(lib().se3_vgtk_ball_query(q.data_ptr(), s.data_ptr(), b, n, m, float(radius), int(n_sample), idx.data_ptr(), _stream()), 'se3_vgtk_ball_query')
    return idx


def anchor_query(sample_idx, grouped_indices, grouped_xyz, anchors, kernel_points, nq=0):
    """grouping.anchor_query (grouping_cuda.cpp:88-108): grouped_xyz (b, 3, p, nn) local coordinates, anchors (na, 3), kernel_points (ks, 2)
    -> [anchor_weights (b, p, na, ks, nn)].  sample_idx / grouped_indices / nq are accepted for signature compatibility (the reference's
    kernel does not read them)."""
    g = _req(grouped_xyz.contiguous(), torch.float32, 'grouped_xyz', 4)
    a, k = _req(anchors.contiguous(), torch.float32, 'anchors', 2), _req(kernel_points.contiguous(), torch.float32, 'kernel_points', 2)
    b, three, p, nn_ = g.shape
    if three != 3 or a.shape[1] != 3 or k.shape[1] != 2:
        raise RuntimeError('anchor_query: grouped_xyz (b, 3, p, nn), anchors (na, 3), kernel_points (ks, 2)')
    out = torch.empty((b, p, a.shape[0], k.shape[0], nn_), dtype=torch.float32, device=g.device)
    check(lib().se3_vgtk_anchor_query(g.data_ptr(), a.data_ptr(), k.data_ptr(), b, p, nn_, a.shape[0], k.shape[0], out.data_ptr(), _stream()),
          'se3_vgtk_anchor_query')
    return [out]


def initial_anchor_query(frag, centers, kernels, r, sigma):
    """vgtk/so3conv/functional.py:108-109: frag (m, 3) points, centers (b, 3, nc), kernels (ks, na, 3) -> [weights, counts] (b, ks, nc, na)."""
    c, x = _req(centers.contiguous(), torch.float32, 'centers', 3), _req(frag.contiguous(), torch.float32, 'frag', 2)
    k = _req(kernels.contiguous(), torch.float32, 'kernels', 3)
    b, three, nc = c.shape
    if three != 3 or x.shape[1] != 3 or k.shape[2] != 3:
        raise RuntimeError('initial_anchor_query: centers (b, 3, nc), frag (m, 3), kernels (ks, na, 3)')
    ks, na = k.shape[0], k.shape[1]
    w = torch.empty((b, ks, nc, na), dtype=torch.float32, device=c.device)
    n = torch.empty_like(w)
    check(lib().se3_vgtk_initial_anchor_query(c.data_ptr(), x.data_ptr(), k.data_ptr(), b, nc, x.shape[0], na, ks, float(r), float(sigma),
                                              w.data_ptr(), n.data_ptr(), _stream()), 'se3_vgtk_initial_anchor_query')
    return [w, n]


def furthest_sample_index(pc, n_sample, lazy_sample=False):
    """pc (b, 3, n) -> (b, n_sample) int32 indices of iterative furthest point sampling starting at point 0."""
    if pc.shape[2] == n_sample or lazy_sample:                      # the reference's shortcut (vgtk/pc/sample.py:64-67)
        return torch.arange(n_sample, device=pc.device).view(1, -1).expand(pc.shape[0], -1).int().contiguous()
    pc = _req(pc.contiguous(), torch.float32, 'pc', 3)
    b, _, n = pc.shape
    temp = torch.empty((b, n), dtype=torch.float32, device=pc.device)
    idx = torch.empty((b, int(n_sample)), dtype=torch.int32, device=pc.device)
    check(lib().se3_vgtk_furthest_point_sampling(pc.data_ptr(), b, n, int(n_sample), temp.data_ptr(), idx.data_ptr(), _stream()),
          'se3_vgtk_furthest_point_sampling')
    return idx


class InterZPConvGrouping(torch.autograd.Function):
    """inter_idx, inter_w (b, np, na, ks, ann), feats (b, c, nq, na) -> (b, c, ks, np, na)."""

    @staticmethod
    def forward(ctx, inter_idx, inter_w, feats):
        idx, w = _req(inter_idx.contiguous(), torch.int32, 'inter_idx', 5), _req(inter_w.contiguous(), torch.float32, 'inter_w', 5)
        f = _req(feats.contiguous(), torch.float32, 'feats', 4)
        b, np_, na, ks, ann = idx.shape
        c, nq = f.shape[1], f.shape[2]
        out = torch.empty((b, c, ks, np_, na), dtype=torch.float32, device=f.device)
        check(lib().se3_vgtk_inter_zpconv_fwd(idx.data_ptr(), w.data_ptr(), f.data_ptr(), b, np_, nq, na, ks, ann, c, out.data_ptr(), _stream()),
              'se3_vgtk_inter_zpconv_fwd')
        ctx.save_for_backward(idx, w)
        ctx.nq = nq
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, w = ctx.saved_tensors
        g = _req(grad_out.contiguous(), torch.float32, 'grad_out', 5)
        b, np_, na, ks, ann = idx.shape
        c = g.shape[1]
        grad = torch.empty((b, c, ctx.nq, na), dtype=torch.float32, device=g.device)
        nbytes = lib().se3_vgtk_inter_zpconv_bwd_workspace_bytes(b, np_, ctx.nq, na, ks, ann)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=g.device)
        check(lib().se3_vgtk_inter_zpconv_bwd(idx.data_ptr(), w.data_ptr(), g.data_ptr(), b, np_, ctx.nq, na, ks, ann, c, grad.data_ptr(),
                                              ws.data_ptr(), nbytes, _stream()), 'se3_vgtk_inter_zpconv_bwd')
        return None, None, grad


def inter_zpconv_grouping(inter_idx, inter_w, feats):
    return InterZPConvGrouping.apply(inter_idx, inter_w, feats)


class IntraZPConvGrouping(torch.autograd.Function):
    """intra_idx (na_out, ann) int32, intra_w (na_out, ks, ann), feats (b, c, np, na_in) -> (b, c, ks, np, na_out)."""

    @staticmethod
    def forward(ctx, intra_idx, intra_w, feats):
        idx, w = _req(intra_idx.contiguous(), torch.int32, 'intra_idx', 2), _req(intra_w.contiguous(), torch.float32, 'intra_w', 3)
        f = _req(feats.contiguous(), torch.float32, 'feats', 4)
        na_out, ann = idx.shape
        ks = w.shape[1]
        b, c, np_, na_in = f.shape
        out = torch.empty((b, c, ks, np_, na_out), dtype=torch.float32, device=f.device)
        check(lib().se3_vgtk_intra_zpconv_fwd(idx.data_ptr(), w.data_ptr(), f.data_ptr(), b, np_, na_in, na_out, ks, ann, c, out.data_ptr(), _stream()),
              'se3_vgtk_intra_zpconv_fwd')
        ctx.save_for_backward(idx, w)
        ctx.na_in = na_in
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, w = ctx.saved_tensors
        g = _req(grad_out.contiguous(), torch.float32, 'grad_out', 5)
        na_out, ann = idx.shape
        b, c, ks, np_, _ = g.shape
        grad = torch.empty((b, c, np_, ctx.na_in), dtype=torch.float32, device=g.device)
        check(lib().se3_vgtk_intra_zpconv_bwd(idx.data_ptr(), w.data_ptr(), g.data_ptr(), b, np_, ctx.na_in, na_out, ks, ann, c, grad.data_ptr(),
                                              _stream()), 'se3_vgtk_intra_zpconv_bwd')
        return None, None, grad


def intra_zpconv_grouping(intra_idx, intra_w, feats):
    return IntraZPConvGrouping.apply(intra_idx, intra_w, feats)


# ---------------------------------------------------------------------------------------------------------------------------------------
# constant tables of the octahedral group under the toolkit's names (host side, numpy / torch; values from se3et_amd.tables)
# ---------------------------------------------------------------------------------------------------------------------------------------
def get_octahedron_vertices():
    """-> vs (6, 3) f32, v_adjs (6, 4) int: the four neighbours of every vertex in ascending order, vRs (24, 3, 3) f32: the group
    rotations, four per vertex, ecs (12, 3) f32: unit edge centres in lexicographic edge order, face_normals (8, 3) f32."""
    from . import tables
    vs = tables.VERTICES.copy()
    d2 = ((vs[:, None] - vs[None]) ** 2).sum(-1)
    v_adjs = np.stack([np.nonzero(d2[i] == 2)[0] for i in range(len(vs))]).astype(np.int64)
    edges = [(i, int(j)) for i in range(len(vs)) for j in v_adjs[i] if j > i]
    ecs = np.stack([vs[i] + vs[j] for i, j in edges]).astype(np.float32)
    ecs /= np.linalg.norm(ecs, axis=1, keepdims=True)
    return vs, v_adjs, tables.rotations().copy(), ecs, tables.face_normals()


def get_relativeV_index(Rs, vs):
    """Permutation of the vertices under the rotations: trace_idx_ori[r, a] = e with R_r v_a = v_e, trace_idx_rot[r, e] = a."""
    moved = np.einsum('rij,aj->rai', np.asarray(Rs, np.float64), np.asarray(vs, np.float64))
    d = ((moved[:, :, None, :] - np.asarray(vs, np.float64)[None, None]) ** 2).sum(-1)
    return d.argmin(2), d.argmin(1)


def label_relative_rotation_simple(anchors, T):
    """The anchor rotation closest to T (largest trace of T anchor^T): -> (T anchor[label]^T (3, 3), label (0-d int64 tensor))."""
    rel = torch.einsum('ij,akj->aik', T, anchors)
    label = rel.diagonal(dim1=-2, dim2=-1).sum(-1).argmax(dim=0)
    return rel[int(label)], label
