"""Deterministic synthetic inputs: random point-cloud pairs of the sizes BASELINE.json names, and
name-keyed synthetic weights (pre-trained SE3ET weights are not released, reference README.md:50-51).

`box_surface` / `make_pair` follow the generator described in SURVEY.md section 8(d): points on the faces of an
axis-aligned box (face chosen with probability proportional to its area) plus Gaussian jitter; the source
cloud is an independently sampled cloud of the same box moved by a fixed rigid transform.
"""
import zlib

import numpy as np

# name -> (n points per cloud, box dims [m], jitter [m])
PAIR_PRESETS = {
    'micro': (600, (0.6, 0.5, 0.4), 0.005),
    'c1_2k': (2000, (1.2, 1.0, 0.8), 0.005),
    'c2_5k': (5000, (1.5, 1.2, 1.0), 0.005),
    'c3_20k': (20000, (60.0, 40.0, 4.0), 0.05),
    'c3_4k': (4000, (30.0, 20.0, 4.0), 0.05),        # KITTI configuration at fixture size
    'cap_30k': (30000, (8.0, 6.0, 4.0), 0.005),      # 3DMatch configuration on a hall: > 2000 superpoints per cloud (the cap)
}


def box_surface(n, dims, seed, jitter):
    g = np.random.default_rng(seed)
    dims = np.asarray(dims, dtype=np.float64)
    areas = np.array([dims[1] * dims[2]] * 2 + [dims[0] * dims[2]] * 2 + [dims[0] * dims[1]] * 2)
    face = g.choice(6, n, p=areas / areas.sum())
    p = g.uniform(0, 1, (n, 3)) * dims
    axis, side = face // 2, face % 2
    p[np.arange(n), axis] = side * dims[axis]
    p += g.normal(0, jitter, (n, 3))
    return p.astype(np.float32)


def euler_zyx(angles):
    """Rotation matrix of intrinsic-free 'zyx' Euler angles as scipy's Rotation.from_euler('zyx', angles)."""
    z, y, x = angles
    cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return Rx @ Ry @ Rz


def make_pair(preset='c2_5k', index=0):
    """Returns (ref (n,3) f32, src (n,3) f32, transform (4,4) f32) with ref ~= src @ R.T + t."""
    n, dims, jitter = PAIR_PRESETS[preset]
    ref = box_surface(n, dims, 2 * index + 1, jitter)
    s0 = box_surface(n, dims, 2 * index + 2, jitter)
    R = euler_zyx([0.5, 0.3, 0.2])
    t = 0.05 * np.asarray(dims)
    src = ((s0 - t) @ R).astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3], T[:3, 3] = R, t
    return ref, src, T


def synth_tensor(name, shape, seed=0):
    """Deterministic float32 values for one named parameter (independent of creation order)."""
    g = np.random.default_rng([seed, zlib.crc32(name.encode())])
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'alpha' or len(shape) == 0:
        return np.float32(1.0) + np.zeros(shape, np.float32)
    if leaf == 'bias':
        return g.uniform(-0.1, 0.1, shape).astype(np.float32)
    if len(shape) == 1:                      # norm scales
        return (1.0 + g.uniform(-0.2, 0.2, shape)).astype(np.float32)
    if leaf == 'weights':                    # KPConvInterSO3 (K_real, A, Cin, Cout): fan-in = slots * Cin
        fan_in = shape[0] * shape[1] * shape[2]
    else:                                    # Linear (out, in)
        fan_in = shape[-1]
    b = np.sqrt(3.0 / fan_in)
    return g.uniform(-b, b, shape).astype(np.float32)
