"""Are two training steps from the same state bit-identical?  Per parameter: identical / largest relative difference of the gradients of two
backward passes (SE3ET-E, one synthetic 5k+5k pair).  python tools/r5/train_determinism.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd.data import registration_collate_fn_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
from se3et_amd.training import OverallLoss
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg), 7).cuda().train()
ref, src, T = make_pair('c2_5k')
d = dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32), src_feats=np.ones((len(src), 1), np.float32), transform=T)
dd = registration_collate_fn_stack_mode([d], cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius, cfg.neighbor_limits)
loss_fn = OverallLoss(cfg)
def grads():
    for p in model.parameters(): p.grad = None
    out = model(dd, train=True, rng=np.random.default_rng(3))      # (the same random selection of ground-truth patch pairs every time)
    loss = loss_fn(out, dd)['loss']
    loss.backward()
    return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
l0, g0 = grads(); l1, g1 = grads(); l2, g2 = grads()
print('losses', l0, l1, l2)
bad = []
for n in g0:
    for o in (g1[n], g2[n]):
        if not torch.equal(o, g0[n]):
            bad.append((float((o - g0[n]).abs().max() / g0[n].abs().max().clamp_min(1e-30)), n)); break
print('%d of %d gradients differ between runs' % (len(bad), len(g0)))
for r, n in sorted(bad, reverse=True)[:12]: print('  %.2e  %s' % (r, n))
