"""Build container only: the C oracle (A1/A2) against the genuine reference extension compiled into oracle/_ref."""
import numpy as np
import pytest
import torch

from helpers import assert_neighbors_equal

pytestmark = pytest.mark.reference


@pytest.mark.parametrize('n1,n2,scale,voxel', [(5000, 4000, 1.0, 0.05), (12000, 9000, 2.0, 0.05), (3000, 100, 0.5, 0.1),
                                               (30, 2, 0.3, 0.1)])
def test_c_oracle_matches_reference_extension(n1, n2, scale, voxel):
    from oracle import native, ref_shims
    ext = ref_shims._RefExt(ref_shims.REF_EXT_SO)
    g = np.random.default_rng(0)
    pts = torch.from_numpy((g.uniform(0, 1, (n1 + n2, 3)) * scale).astype(np.float32))
    nrm = torch.from_numpy(g.normal(size=(n1 + n2, 3)).astype(np.float32))
    lens = torch.tensor([n1, n2])
    a = ext.grid_subsampling(pts, lens, nrm, voxel)
    b = native.grid_subsample(pts, lens, nrm, voxel)
    assert a[1].tolist() == b[1].tolist()
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])          # same points, same unordered_map order
    sp, sl = a[0], a[1]
    na = ext.radius_neighbors(sp, sp, sl, sl, voxel * 2.5)[:, :38]
    nb = native.radius_search(sp, sp, sl, sl, voxel * 2.5, 38)
    assert_neighbors_equal(nb, na, sp, sp, 'radius')
