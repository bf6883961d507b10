"""Rigid-transform helpers the callers of the hot path import next to the operators (geotransformer/modules/ops/transformation.py:
apply_transform :7-60, apply_rotation :63-107, get_rotation_translation_from_transform :110-122, get_transform_from_rotation_translation
:125-143, inverse_transform :146-159).  Plain torch on whatever device the tensors live on: a 3 x 3 product per point is not a kernel.
The Rodrigues helpers and vector_angle of the reference's package are not mirrored; se3et_amd.dropin resolves them from the reference
tree when it is importable."""
import torch


def _rigid(points, matrix, offset, normals, what):
    if normals is not None and normals.shape != points.shape:
        raise AssertionError('points %s and normals %s differ in shape' % (tuple(points.shape), tuple(normals.shape)))
    if matrix.ndim == 2:                                    # one transform for every point, any leading shape
        flat = lambda t: (t.reshape(-1, 3) @ matrix.t()).reshape(t.shape)
        moved = flat(points) if offset is None else (points.reshape(-1, 3) @ matrix.t() + offset).reshape(points.shape)
    elif matrix.ndim == 3 and points.ndim == 3:             # one transform per batch entry (points broadcast over B = 1)
        flat = lambda t: t @ matrix.transpose(-1, -2)
        moved = flat(points) if offset is None else flat(points) + offset[:, None, :]
    else:
        raise ValueError('Incompatible shapes between points {} and {} {}.'.format(tuple(points.shape), what, tuple(matrix.shape)))
    return moved if normals is None else (moved, flat(normals))


def apply_transform(points, transform, normals=None):
    """Q = P R^T + t for (*, 3) points and a (4, 4) transform, or (B, N, 3) points and (B, 4, 4) transforms; normals only rotate."""
    return _rigid(points, transform[..., :3, :3], transform[..., :3, 3], normals, 'transform')


def apply_rotation(points, rotation, normals=None):
    return _rigid(points, rotation, None, normals, 'rotation')


def get_rotation_translation_from_transform(transform):
    return transform[..., :3, :3], transform[..., :3, 3]


def get_transform_from_rotation_translation(rotation, translation):
    out = torch.zeros(rotation.shape[:-2] + (4, 4), dtype=rotation.dtype, device=rotation.device)
    out[..., :3, :3], out[..., :3, 3], out[..., 3, 3] = rotation, translation, 1.0
    return out


def inverse_transform(transform):
    r, t = get_rotation_translation_from_transform(transform)
    rt = r.transpose(-1, -2)
    return get_transform_from_rotation_translation(rt, -(rt @ t.unsqueeze(-1)).squeeze(-1))
