"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/c/grid_radius_oracle.c (rows A1/A2 of SURVEY.md
section 8a).  Signatures follow the reference wrappers geotransformer/modules/ops/grid_subsample.py:7-24 and
radius_search.py:7-27 (CPU float32 / int64 tensors in, fresh tensors out)."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'c', 'grid_radius_oracle.c')
_SO = os.path.join(_HERE, '_build', 'liboracle_c.so')
_lib = None


def build(force=False):
    """gcc the C restatement (no FMA contraction, no -march: the float expressions must stay unfused)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fPIC', '-shared', '-o', _SO, _SRC, '-lm'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, i64, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
        L.oracle_grid_subsample.restype = i64
        L.oracle_grid_subsample.argtypes = [vp, vp, i64, vp, i64, f32, vp, vp, vp]
        L.oracle_radius_neighbors.restype = i64
        L.oracle_radius_neighbors.argtypes = [vp, i64, vp, i64, vp, vp, i64, f32, i64, vp]
        _lib = L
    return _lib


def _check(t, dtype, name):
    if t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError('%s must be a contiguous CPU %s tensor' % (name, dtype))


def grid_subsample(points, lengths, normals, voxel_size):
    _check(points, torch.float32, 'points')
    _check(normals, torch.float32, 'normals')
    _check(lengths, torch.int64, 'lengths')
    n = points.shape[0]
    s_points = torch.empty((n, 3), dtype=torch.float32)
    s_normals = torch.empty((n, 3), dtype=torch.float32)
    s_lengths = torch.empty_like(lengths)
    m = lib().oracle_grid_subsample(points.data_ptr(), normals.data_ptr(), n, lengths.data_ptr(),
                                    lengths.shape[0], float(voxel_size), s_points.data_ptr(),
                                    s_normals.data_ptr(), s_lengths.data_ptr())
    return s_points[:m].clone(), s_lengths, s_normals[:m].clone()


def radius_search(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit):
    _check(q_points, torch.float32, 'q_points')
    _check(s_points, torch.float32, 's_points')
    _check(q_lengths, torch.int64, 'q_lengths')
    _check(s_lengths, torch.int64, 's_lengths')
    nq, ns = q_points.shape[0], s_points.shape[0]
    cap = int(neighbor_limit) if neighbor_limit > 0 else max(int(s_lengths.max()), 1)
    out = torch.empty((nq, cap), dtype=torch.int64)
    max_count = lib().oracle_radius_neighbors(q_points.data_ptr(), nq, s_points.data_ptr(), ns,
                                              q_lengths.data_ptr(), s_lengths.data_ptr(), q_lengths.shape[0],
                                              float(radius), cap, out.data_ptr())
    return out[:, :min(cap, max_count)].contiguous()
