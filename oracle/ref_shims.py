"""TEST INFRASTRUCTURE ONLY -- import shims that let the *genuine* reference
(/root/reference, a read-only Python tree) be imported on a CPU-only container.

Used by ``tests/golden/generate_golden.py`` (fixture generator) and by the
oracle-vs-reference validation tests, and by nothing else: nothing in
``se3et_amd`` (the product), ``bench.py`` or the ``-m gpu`` tests may import
this module, and it is a no-op on the GPU box where /root/reference is absent.

What is shimmed (recipe: SURVEY.md Appendix A):
  * third-party modules that are not installed here and are not on the hot
    path: IPython, ipdb, coloredlogs, turtle, easydict, open3d, plyfile;
  * ``vgtk.cuda.{zpconv,gathering,grouping}``: empty modules (import-time only,
    the SE3ET models never call them);
  * ``trimesh``: a small stand-in (binary PLY reader, face normals, edge and
    adjacency tables) used only to derive the constant anchor tables;
  * ``e3nn.o3``: closed forms for l <= 1 (Y0 = 1/(2 sqrt(pi)); Y1 = sqrt(3/4pi)
    * unit vector in (x, y, z) order; D^0 = 1; D^1(R) = R).  e3nn itself is not
    installable here, so this convention is *unpinned* (DESIGN.md);
  * ``geotransformer.ext``: the reference C++ sources compiled as they lie
    (oracle/ref_ext/Makefile -> oracle/_ref/libref_ext.so) and wrapped with
    ctypes to the pybind signature;
  * ``Tensor.cuda()/Module.cuda()`` -> identity, so the hard-coded ``.cuda()``
    calls of the reference run on CPU.
"""
import ctypes
import logging
import os
import struct
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get('SE3ET_REFERENCE_ROOT', '/root/reference')
_HERE = os.path.dirname(os.path.abspath(__file__))
REF_EXT_SO = os.path.join(_HERE, '_ref', 'libref_ext.so')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'geotransformer'))


# ----------------------------------------------------------------------------
# small stubs
# ----------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


# ----------------------------------------------------------------------------
# trimesh stand-in
# ----------------------------------------------------------------------------
class _Trimesh:
    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)

    def fix_normals(self):
        c = self.vertices.mean(0)
        f = self.faces.copy()
        for i, (a, b, d) in enumerate(f):
            n = np.cross(self.vertices[b] - self.vertices[a], self.vertices[d] - self.vertices[a])
            if np.dot(n, self.vertices[[a, b, d]].mean(0) - c) < 0:
                f[i] = [a, d, b]
        self.faces = f

    @property
    def face_normals(self):
        v = self.vertices
        n = np.cross(v[self.faces[:, 1]] - v[self.faces[:, 0]], v[self.faces[:, 2]] - v[self.faces[:, 0]])
        return n / np.linalg.norm(n, axis=1, keepdims=True)

    @property
    def edges(self):
        return self.faces[:, [0, 1, 1, 2, 2, 0]].reshape(-1, 2)

    @property
    def edges_unique(self):
        e = np.sort(self.edges, axis=1)
        return np.unique(e, axis=0)

    @property
    def vertex_neighbors(self):
        nb = [[] for _ in range(len(self.vertices))]
        for a, b in self.edges:
            if b not in nb[a]:
                nb[a].append(int(b))
            if a not in nb[b]:
                nb[b].append(int(a))
        return nb

    @property
    def face_adjacency(self):
        edge_faces = {}
        for fi, f in enumerate(self.faces):
            for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
                edge_faces.setdefault((min(a, b), max(a, b)), []).append(fi)
        pairs = [sorted(v) for v in edge_faces.values() if len(v) == 2]
        return np.array(sorted(pairs), dtype=np.int64)


def _load_ply(path):
    with open(path, 'rb') as f:
        data = f.read()
    end = data.index(b'end_header\n') + len(b'end_header\n')
    header = data[:end].decode('ascii').splitlines()
    nv = nf = 0
    for line in header:
        if line.startswith('element vertex'):
            nv = int(line.split()[-1])
        if line.startswith('element face'):
            nf = int(line.split()[-1])
    off = end
    verts = []
    for _ in range(nv):
        x, y, z = struct.unpack_from('<fff', data, off)
        off += 12 + 4
        verts.append((x, y, z))
    faces = []
    for _ in range(nf):
        n = data[off]
        off += 1
        idx = struct.unpack_from('<' + 'i' * n, data, off)
        off += 4 * n
        nt = data[off]
        off += 1 + 4 * nt
        off += 4
        faces.append(idx)
    return _Trimesh(np.array(verts), np.array(faces))


# ----------------------------------------------------------------------------
# e3nn.o3 stand-in, l <= 1 (convention documented in the module docstring)
# ----------------------------------------------------------------------------
class _Irrep:
    def __init__(self, l, p):
        assert l in (0, 1) and p == 1
        self.l = l

    def D_from_matrix(self, R):
        if self.l == 0:
            return torch.ones(R.shape[:-2] + (1, 1), dtype=R.dtype)
        return R.clone()


def _spherical_harmonics(ls, x, normalize, normalization='integral'):
    assert normalize and normalization == 'integral'
    out = []
    for l in ls:
        if l == 0:
            out.append(torch.full(x.shape[:-1] + (1,), 0.5 / np.sqrt(np.pi), dtype=x.dtype))
        elif l == 1:
            out.append(np.sqrt(3.0 / (4.0 * np.pi)) * torch.nn.functional.normalize(x, dim=-1))
        else:
            raise NotImplementedError(l)
    return torch.cat(out, -1)


# ----------------------------------------------------------------------------
# geotransformer.ext over the compiled reference sources
# ----------------------------------------------------------------------------
class _RefExt:
    def __init__(self, so_path):
        self.lib = ctypes.CDLL(so_path)
        L = self.lib
        L.ref_radius_neighbors.restype = ctypes.c_long
        L.ref_radius_neighbors.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_float]
        L.ref_fetch_neighbors.argtypes = [ctypes.c_void_p]
        L.ref_grid_subsampling.restype = ctypes.c_long
        L.ref_grid_subsampling.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p,
                                           ctypes.c_long, ctypes.c_float]
        L.ref_fetch_subsampled.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    @staticmethod
    def _chk(t, dtype):
        if t.is_cuda or t.dtype != dtype or not t.is_contiguous():
            raise RuntimeError('geotransformer.ext: expected a contiguous CPU %s tensor' % dtype)

    def radius_neighbors(self, q_points, s_points, q_lengths, s_lengths, radius):
        for t in (q_points, s_points):
            self._chk(t, torch.float32)
        for t in (q_lengths, s_lengths):
            self._chk(t, torch.int64)
        nq = q_points.shape[0]
        width = self.lib.ref_radius_neighbors(q_points.data_ptr(), nq, s_points.data_ptr(), s_points.shape[0],
                                              q_lengths.data_ptr(), s_lengths.data_ptr(), q_lengths.shape[0],
                                              float(radius))
        out = torch.zeros((nq, width), dtype=torch.int64)
        self.lib.ref_fetch_neighbors(out.data_ptr())
        return out

    def grid_subsampling(self, points, lengths, normals, voxel_size):
        self._chk(points, torch.float32)
        self._chk(normals, torch.float32)
        self._chk(lengths, torch.int64)
        m = self.lib.ref_grid_subsampling(points.data_ptr(), normals.data_ptr(), points.shape[0],
                                          lengths.data_ptr(), lengths.shape[0], float(voxel_size))
        s_points = torch.zeros((m, 3), dtype=torch.float32)
        s_normals = torch.zeros((m, 3), dtype=torch.float32)
        s_lengths = torch.zeros((lengths.shape[0],), dtype=torch.int64)
        self.lib.ref_fetch_subsampled(s_points.data_ptr(), s_normals.data_ptr(), s_lengths.data_ptr())
        return [s_points, s_lengths, s_normals]


_INSTALLED = False


def install(variant='se3ete.3dmatch'):
    """Install every shim and put the reference on sys.path.  Idempotent."""
    global _INSTALLED
    if not reference_available():
        raise RuntimeError('reference tree not found at %s' % REFERENCE_ROOT)
    exp_dir = os.path.join(REFERENCE_ROOT, 'experiments', variant)
    if _INSTALLED:
        return
    _mod('IPython', embed=lambda *a, **k: None)
    _mod('ipdb', set_trace=lambda *a, **k: None)
    _mod('coloredlogs', ColoredFormatter=logging.Formatter)
    _mod('turtle', forward=lambda *a, **k: None)
    _mod('easydict', EasyDict=_EasyDict)
    _mod('open3d')
    _mod('plyfile', PlyElement=object, PlyData=object)
    tm = _mod('trimesh', Trimesh=_Trimesh, load=_load_ply)
    tm.base = _mod('trimesh.base', Trimesh=_Trimesh)
    e3 = _mod('e3nn')
    e3.o3 = _mod('e3nn.o3', Irrep=_Irrep, spherical_harmonics=_spherical_harmonics)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    torch.cuda.synchronize = lambda *a, **k: None

    vg = os.path.join(REFERENCE_ROOT, 'geotransformer', 'modules', 'e2pn', 'vgtk')
    for p in (REFERENCE_ROOT, vg):
        if p not in sys.path:
            sys.path.insert(0, p)
    # vgtk.cuda.* : import-time only (found in sys.modules before the real package is searched)
    cuda_pkg = _mod('vgtk.cuda')
    cuda_pkg.__path__ = []
    for sub in ('zpconv', 'gathering', 'grouping'):
        setattr(cuda_pkg, sub, _mod('vgtk.cuda.' + sub))

    if not os.path.exists(REF_EXT_SO):
        raise RuntimeError('build oracle/_ref first: make -C oracle/ref_ext')
    import geotransformer  # noqa: F401  (namespace of the reference)
    ext = _RefExt(REF_EXT_SO)
    m = _mod('geotransformer.ext', radius_neighbors=ext.radius_neighbors, grid_subsampling=ext.grid_subsampling)
    sys.modules['geotransformer'].ext = m
    import vgtk
    vgtk.cuda = cuda_pkg

    import geotransformer.utils.common as common
    common.ensure_dir = lambda *a, **k: None
    import geotransformer.modules.geotransformer  # noqa: F401  must precede .transformer
    import geotransformer.utils.open3d as o3d_utils
    zeros = lambda pts, *a, **k: np.zeros((len(pts), 3), dtype=np.float64)
    o3d_utils.estimate_normals = zeros
    import geotransformer.utils.data as data_utils
    data_utils.estimate_normals = zeros
    if exp_dir not in sys.path:
        sys.path.insert(0, exp_dir)
    _INSTALLED = True


def load_experiment(variant):
    """Return (make_cfg, create_model) of one reference experiment directory, isolated from the
    others (they all use the top-level module names config/model/backbone)."""
    install(variant)
    exp_dir = os.path.join(REFERENCE_ROOT, 'experiments', variant)
    for name in ('config', 'model', 'backbone', 'loss', 'dataset'):
        sys.modules.pop(name, None)
    sys.path = [p for p in sys.path if not p.startswith(os.path.join(REFERENCE_ROOT, 'experiments'))]
    sys.path.insert(0, exp_dir)
    import config as cfg_mod
    import model as model_mod
    return cfg_mod.make_cfg, model_mod.create_model
