"""Drop-in for the reference's pybind module `geotransformer.ext` on HOST memory (extensions/pybind.cpp:6-18): the two functions the
collate function calls inside forked DataLoader workers (geotransformer/utils/data.py:159-209), where a GPU context is off limits.
Same signatures, same checks (CPU, contiguous, float32 / int64 -> RuntimeError as the reference's TORCH_CHECKs,
extensions/common/torch_helper.h:6-35), same results bit for bit (tests/test_host_ext.py).  The GPU pipeline
(se3et_amd.data.precompute_data_stack_mode) does not use this module."""
import ctypes

import torch

from ._lib import check, lib


def _chk(t, dtype, name):
    if not torch.is_tensor(t) or t.is_cuda:
        raise RuntimeError('%s must be a CPU tensor' % name)
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)
    if t.dtype != dtype:
        raise RuntimeError('%s must be a %s tensor' % (name, str(dtype).replace('torch.', '')))
    return t


def grid_subsampling(points, lengths, normals, voxel_size):
    """-> [s_points (M, 3), s_lengths (B,), s_normals (M, 3)] (grid_subsampling.cpp:5-83)."""
    _chk(points, torch.float32, 'points'), _chk(lengths, torch.int64, 'lengths'), _chk(normals, torch.float32, 'normals')
    n, b = points.shape[0], lengths.shape[0]
    s_points, s_normals = torch.empty((n, 3), dtype=torch.float32), torch.empty((n, 3), dtype=torch.float32)
    s_lengths = torch.empty((b,), dtype=torch.int64)
    check(lib().se3_grid_subsample_host(points.data_ptr(), normals.data_ptr(), n, lengths.data_ptr(), b, float(voxel_size), s_points.data_ptr(),
                                        s_normals.data_ptr(), s_lengths.data_ptr()), 'se3_grid_subsample_host')
    m = int(s_lengths.sum())
    return [s_points[:m].clone(), s_lengths, s_normals[:m].clone()]


def radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius):
    """-> LongTensor (Nq, max_count), rows ascending in distance, padded with Ns (radius_neighbors.cpp:5-76)."""
    _chk(q_points, torch.float32, 'q_points'), _chk(s_points, torch.float32, 's_points')
    _chk(q_lengths, torch.int64, 'q_lengths'), _chk(s_lengths, torch.int64, 's_lengths')
    nq, ns, b = q_points.shape[0], s_points.shape[0], q_lengths.shape[0]
    mc = ctypes.c_int64(0)
    args = (q_points.data_ptr(), nq, s_points.data_ptr(), ns, q_lengths.data_ptr(), s_lengths.data_ptr(), b, float(radius))
    check(lib().se3_radius_neighbors_host(*args, 0, None, ctypes.byref(mc)), 'se3_radius_neighbors_host')
    width = max(int(mc.value), 0)
    out = torch.empty((nq, width), dtype=torch.int64)
    if width > 0:
        check(lib().se3_radius_neighbors_host(*args, width, out.data_ptr(), ctypes.byref(mc)), 'se3_radius_neighbors_host')
    return out
