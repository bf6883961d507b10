"""GPU: equivariance self-check of the HIP path under the 24 group rotations (property test, no oracle involved): see
tests/equivariance.py for what is checked and why the weights are projected onto the exactly-equivariant subspace first."""
import numpy as np
import pytest
import torch

from equivariance import group, rotated_data, symmetrise_state
from helpers import assert_close

pytestmark = pytest.mark.gpu


def _symmetric_model(variant):
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    cfg = make_cfg(variant)
    model = load_synthetic_weights(create_model(cfg), 3)
    model.load_state_dict(symmetrise_state(model.state_dict()), strict=True)
    return cfg, model.cuda().eval()


def _pyramid(cfg, preset, indices):
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.synthetic import make_pair
    clouds = []
    for j in indices:
        ref, src, _ = make_pair(preset, index=j)
        clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
    b = cfg.backbone
    data = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius,
                                      cfg.neighbor_limits)
    data['features'] = torch.ones((pts.shape[0], 1), device='cuda')
    return data


def _run(model, data):
    taps = {}
    model.transformer.transformer.layer_tap = lambda i, t: taps.__setitem__(i, t[0])
    out = model(data, with_registration=False)
    return out, taps


@pytest.mark.parametrize('variant,preset,rotations', [('micro_e', 'micro', range(24)), ('micro_i', 'micro', range(24)),
                                                      ('se3ete', 'c1_2k', (1, 6, 13, 20)), ('se3eti', 'c1_2k', (3, 10, 17))])
def test_forward_is_equivariant_on_the_symmetric_weight_subspace(variant, preset, rotations):
    cfg, model = _symmetric_model(variant)
    data = _pyramid(cfg, preset, [0])
    Rs, perms = group()
    base, taps0 = _run(model, data)
    blocks = cfg.geotransformer.blocks
    for g in rotations:
        out, taps = _run(model, rotated_data(data, Rs[g]))
        p = perms[g].cuda()
        assert_close(out['feats_c'], base['feats_c'][:, p], 1e-4, 'rotation %d: backbone feats_c' % g)
        assert_close(out['feats_f'], base['feats_f'], 1e-4, 'rotation %d: fine features' % g)
        for i, block in enumerate(blocks):
            want = taps0[i][p] if taps0[i].dim() == 3 else taps0[i]
            assert_close(taps[i], want, 2e-4, 'rotation %d: layer %d (%s)' % (g, i, block))
        assert_close(out['ref_feats_c'], base['ref_feats_c'], 2e-4, 'rotation %d: ref_feats_c' % g)
        assert_close(out['src_feats_c'], base['src_feats_c'], 2e-4, 'rotation %d: src_feats_c' % g)


def test_several_pairs_per_forward_are_equivariant_too():
    """The stacked forward (se3et_amd.batched) on two pairs: per-pair results follow the single-pair property."""
    from se3et_amd.batched import forward_pairs
    cfg, model = _symmetric_model('micro_e')
    data = _pyramid(cfg, 'micro', [0, 1])
    Rs, perms = group()
    base = forward_pairs(model, data, with_registration=False)
    for g in (4, 11, 22):
        outs = forward_pairs(model, rotated_data(data, Rs[g]), with_registration=False)
        for o, w in zip(outs, base):
            assert_close(o['feats_c'], w['feats_c'][:, perms[g].cuda()], 1e-4, 'rotation %d feats_c' % g)
            assert_close(o['ref_feats_c'], w['ref_feats_c'], 2e-4, 'rotation %d ref_feats_c' % g)
            assert_close(o['src_feats_c'], w['src_feats_c'], 2e-4, 'rotation %d src_feats_c' % g)
