"""Tensor-level front end of the C ABI: validates tensors the way the reference extension does with TORCH_CHECK
(device / dtype / contiguity -> RuntimeError), allocates outputs with torch (device memory stays owned by
PyTorch) and launches the HIP kernels on the current torch stream."""
import ctypes

import torch

from ._lib import check, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype, name, ndim=None):
    if not torch.is_tensor(t):
        raise RuntimeError('%s must be a tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must be a GPU tensor (the SE3ET hot path has no CPU implementation)' % name)
    if t.dtype != dtype:
        raise RuntimeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError('%s must have %d dims, got %d' % (name, ndim, t.dim()))
    return t


def _host_lengths(lengths, name):
    """Batch lengths live on the host (as in the reference, where they are CPU LongTensors)."""
    if torch.is_tensor(lengths):
        if lengths.dtype != torch.int64:
            raise RuntimeError('%s must be int64' % name)
        lengths = lengths.tolist()
    arr = (ctypes.c_int64 * len(lengths))(*[int(v) for v in lengths])
    return arr, len(lengths)


def radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius, limit):
    """Returns (neighbors (Nq, limit) int64 padded with Ns, max_count 0-d int32 device tensor)."""
    _req(q_points, torch.float32, 'q_points', 2)
    _req(s_points, torch.float32, 's_points', 2)
    ql, nb = _host_lengths(q_lengths, 'q_lengths')
    sl, nb2 = _host_lengths(s_lengths, 's_lengths')
    if nb != nb2:
        raise RuntimeError('q_lengths and s_lengths differ in batch size')
    nq, ns = q_points.shape[0], s_points.shape[0]
    out = torch.empty((nq, limit), dtype=torch.int64, device=q_points.device)
    max_count = torch.empty((), dtype=torch.int32, device=q_points.device)
    check(lib().se3_radius_neighbors(q_points.data_ptr(), nq, s_points.data_ptr(), ns, ql, sl, nb, float(radius),
                                     int(limit), out.data_ptr(), max_count.data_ptr(), _stream()),
          'se3_radius_neighbors')
    return out, max_count


def grid_subsample(points, lengths, normals, voxel_size):
    """Returns (s_points (N,3) [first sum(s_lengths) rows valid], s_normals or None, s_lengths (B,) int64 device)."""
    _req(points, torch.float32, 'points', 2)
    if normals is not None:
        _req(normals, torch.float32, 'normals', 2)
    ln, nb = _host_lengths(lengths, 'lengths')
    n = points.shape[0]
    dev = points.device
    ws_bytes = lib().se3_grid_subsample_workspace_bytes(n, nb)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    s_points = torch.empty((n, 3), dtype=torch.float32, device=dev)
    s_normals = torch.empty((n, 3), dtype=torch.float32, device=dev) if normals is not None else None
    s_lengths = torch.empty((nb,), dtype=torch.int64, device=dev)
    check(lib().se3_grid_subsample(points.data_ptr(), normals.data_ptr() if normals is not None else None, n, ln, nb,
                                   float(voxel_size), s_points.data_ptr(),
                                   s_normals.data_ptr() if s_normals is not None else None, s_lengths.data_ptr(),
                                   ws.data_ptr(), ws_bytes, _stream()), 'se3_grid_subsample')
    return s_points, s_normals, s_lengths
