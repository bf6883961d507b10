"""CPU: the C-ABI library loads and exports every symbol include/se3et_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest


def _declared():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'include', 'se3et_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(se3_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from se3et_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), 'missing symbol %s' % n
        assert n in _lib.SIGNATURES, 'no ctypes signature for %s' % n
    assert set(_lib.SIGNATURES) == set(names)
    assert b'gfx950' in L.se3_version()


def test_plan_structs_match_the_library():
    """The ctypes mirror of se3_linear_t / se3_layer_t / se3_transformer_plan_t (se3et_amd/cdriver.py) has the library's sizes and offsets."""
    from se3et_amd import cdriver
    cdriver.check_layout()
    assert ctypes.sizeof(cdriver.Plan) > 16 * ctypes.sizeof(cdriver.Layer)


def test_argument_validation_without_gpu():
    from se3et_amd import _lib
    L = _lib.lib()
    # invalid arguments are rejected on the host before any launch
    assert L.se3_radius_neighbors(None, 0, None, 0, None, None, 0, 0.1, 10, None, None, None) != 0
    assert b'radius_neighbors' in L.se3_last_error()
    assert L.se3_grid_subsample_workspace_bytes(10000, 2) > 0
    assert L.se3_group_norm_workspace_bytes(1000, 64, 32) > 0
    assert L.se3_attention_fwd(None, None, None, None, 1, 1, 1, 32, 4, 32, 32, 32, 0, 0, 0, 0, 0, 1.0, None, None) != 0


def test_product_refuses_cpu_tensors():
    import torch
    from se3et_amd.modules.ops import radius_search
    with pytest.raises(RuntimeError):
        radius_search(torch.zeros(4, 3), torch.zeros(4, 3), torch.tensor([4]), torch.tensor([4]), 0.1, 8)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from se3et_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU / eager fallback'):
        _lib.lib()


def test_cdriver_supported_mirrors_the_drivers_scheduler():
    """cdriver.supported() must refuse every block list se3_transformer_forward would refuse (csrc/transformer_driver.hip:
    SE3_REQUIREs on the X_is_eq / Xeq state), so that such models keep the Python schedule instead of raising."""
    from se3et_amd.cdriver import _schedule_ok
    e = ['self_eq', 'cross_a_soft', 'self_eq', 'cross_r_soft', 'self', 'cross', 'self', 'cross', 'self', 'cross']
    i = ['self_eq', 'cross', 'self_eq', 'cross', 'self_eq', 'cross']
    assert _schedule_ok(e, True) and _schedule_ok(i, True)
    assert not _schedule_ok(e, False)                                   # eq2inv without the rotcompress layer
    assert not _schedule_ok(['self_eq', 'cross_a_soft'], True)          # ends on anchor features
    assert not _schedule_ok(['self_eq', 'cross_r_soft'], True)
    assert not _schedule_ok(['self_eq', 'self', 'cross'], True)         # plain cross attention on anchor features
    assert not _schedule_ok(['self_eq', 'cross', 'cross_a_soft'], True)  # equivariant cross attention on invariant features
    assert _schedule_ok(['self_eq', 'cross_r_soft', 'self', 'cross'], True)
