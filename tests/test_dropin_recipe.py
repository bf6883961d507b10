"""Build container only (needs /root/reference): INTEGRATION.md section 1 executed verbatim against the reference's own experiment
directories.  The recipe text is read out of INTEGRATION.md by tests/dropin_probe.py, so the documentation and
this test cannot drift; every experiment runs in a fresh interpreter because the aliases rewrite sys.modules."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXPERIMENTS = ['se3ete.3dmatch', 'se3ete2.3dmatch', 'se3eti.3dmatch', 'se3eti.kitti', 'se3eti2.3dmatch']


@pytest.mark.reference
@pytest.mark.parametrize('experiment', EXPERIMENTS)
def test_integration_recipe_builds_the_reference_experiment(experiment):
    env = dict(os.environ, SE3_BLOCKING_SYNC='0')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'dropin_probe.py'), experiment], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['foreign_modules'] == [], 'operator modules not from se3et_amd: %s' % out['foreign_modules']
    assert out['model_class_module'] == 'model'                                   # the reference's own GeoTransformer class ...
    assert out['transformer_module'].startswith('se3et_amd.modules.geotransformer')        # ... built on this package's operators
    assert out['target_generator_module'] == 'se3et_amd.modules.geotransformer.superpoint_target'
    assert out['node_correspondences_module'] == 'se3et_amd.modules.registration.matching'
    assert out['point_to_node_partition_module'] == 'se3et_amd.modules.ops.pointcloud_partition'
    assert out['target_rng_matches'], 'SuperPointTargetGenerator does not draw like superpoint_target.py:30-35'
    assert out['only_reference'] == [] and out['only_ours'] == [] and out['mismatched'] == [], out
    assert out['n_keys'] > 300
    assert '/reference/' in out['metrics_from'] or 'geotransformer/modules/registration/metrics.py' in out['metrics_from']


def test_recipe_text_is_the_helper_call():
    """The documented block is the two-line call of se3et_amd.dropin (no hand-maintained module list in the text)."""
    sys.path.insert(0, HERE)
    import dropin_probe
    lines = [l for l in dropin_probe.recipe_text().strip().splitlines() if l.strip()]
    assert lines == ['import se3et_amd.dropin', 'se3et_amd.dropin.install_aliases()']


def test_aliases_without_a_reference_tree():
    """On a machine without the reference (the GPU box): the aliases install, the mirrored names import under the reference's module
    names, names that are not mirrored raise AttributeError / ImportError instead of resolving to anything else."""
    code = '''
import sys
import se3et_amd.dropin as d
sys.path = [p for p in sys.path if "reference" not in p]
names = d.install_aliases()
from geotransformer.modules.geotransformer import GeometricTransformer, SuperPointMatching, SuperPointTargetGenerator, LocalGlobalRegistration
from geotransformer.modules.registration import get_node_correspondences
from geotransformer.modules.ops import point_to_node_partition, index_select, apply_transform, pairwise_distance
from geotransformer.modules.ops.transformation import apply_transform as at2
from geotransformer.modules.ops.pairwise_distance import pairwise_distance as pd2
from geotransformer.modules.sinkhorn import LearnableLogOptimalTransport
from geotransformer.modules.kpconv import UnaryBlock, LastUnaryBlock, nearest_upsample
from geotransformer.modules.e2pn.blocks_epn import LiftBlockEPN, SimpleBlockEPN, ResnetBottleneckBlockEPN, UnaryBlockEPN, LastUnaryBlockEPN, InvOutBlockEPN
from geotransformer.modules.transformer.rotation_supervision import RotationAttentionLayer
from geotransformer.modules.transformer.permutation_invariant import PermutationInvariantLayer
import vgtk.functional as fr, vgtk.so3conv as sptk
import geotransformer.modules.ops as ops
try:
    ops.rodrigues_rotation_matrix
    raise SystemExit("unmirrored name resolved without a reference tree")
except AttributeError:
    pass
import torch
T = torch.eye(4); T[:3, 3] = torch.tensor([1., 2., 3.])
p = torch.arange(12.).reshape(4, 3)
assert torch.equal(apply_transform(p, T), p + T[:3, 3])
assert torch.allclose(ops.inverse_transform(T) @ T, torch.eye(4))
print("ok", len(names))
'''
    env = dict(os.environ, SE3_BLOCKING_SYNC='0', PYTHONPATH=os.path.dirname(HERE))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600, cwd='/tmp')
    assert r.returncode == 0 and r.stdout.strip().startswith('ok'), r.stderr[-3000:]


@pytest.mark.reference
def test_toolkit_constants_match_the_reference():
    """get_octahedron_vertices / get_relativeV_index / label_relative_rotation_simple of se3et_amd.vgtk (what the reference's loss.py:88-145
    takes from the EPN toolkit at construction time) against the genuine toolkit imported through oracle/ref_shims (its trimesh is a
    stand-in there: the edge centres are compared as a set)."""
    code = '''
import numpy as np, torch
from oracle import ref_shims
ref_shims.install()
import vgtk.so3conv as sptk, vgtk.functional as fr
import se3et_amd.vgtk as own
a, b = sptk.get_octahedron_vertices(), own.get_octahedron_vertices()
for i, name in ((0, "vs"), (1, "v_adjs"), (2, "vRs"), (4, "face_normals")):
    assert np.asarray(a[i]).shape == np.asarray(b[i]).shape and np.allclose(a[i], b[i], atol=1e-6), name
key = lambda e: sorted(map(tuple, np.round(np.asarray(e, np.float64), 5).tolist()))
assert key(a[3]) == key(b[3]), "ecs"
ta, tb = fr.get_relativeV_index(a[2], a[0]), own.get_relativeV_index(b[2], b[0])
assert np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1], tb[1])
g = np.random.default_rng(0)
anchors = torch.tensor(a[2], dtype=torch.float32)
for _ in range(20):
    q, _r = np.linalg.qr(g.normal(size=(3, 3)))
    q = torch.tensor(q * np.sign(np.linalg.det(q)), dtype=torch.float32)
    ra, la = fr.label_relative_rotation_simple(anchors, q)
    rb, lb = own.label_relative_rotation_simple(anchors, q)
    assert int(la) == int(lb) and torch.allclose(ra, rb)
print("ok")
'''
    env = dict(os.environ, SE3_BLOCKING_SYNC='0', PYTHONPATH=os.path.dirname(HERE))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600, cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-3000:]
