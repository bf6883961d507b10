import torch

from ... import ops as _ops


def grid_subsample(points, lengths, normals, voxel_size):
    """Grid subsampling in stack mode, on the GPU.

    Same contract as the reference wrapper (geotransformer/modules/ops/grid_subsample.py:7-24): per voxel the
    input point nearest to the voxel mean, in the reference's emission order.  `points` / `normals` are GPU float32
    tensors, `lengths` a (host or device) int64 tensor; returns (s_points, s_lengths [host int64], s_normals).
    One host synchronisation (the per-cloud counts size the outputs)."""
    s_points, s_normals, s_lengths = _ops.grid_subsample(points, lengths, normals, voxel_size)
    s_lengths = s_lengths.cpu()
    m = int(s_lengths.sum())
    return s_points[:m], s_lengths, (s_normals[:m] if s_normals is not None else None)
