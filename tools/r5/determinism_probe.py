"""Which pieces of the training step are not run-to-run deterministic?  (a) library GEMMs of the dW / dX shapes, repeated; (b) torch ops that
warn under torch.use_deterministic_algorithms(warn_only=True) during one training step.  python tools/r5/determinism_probe.py"""
import os, sys, warnings; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd import ops
g = torch.Generator(device='cpu').manual_seed(0)
for (m, k, n) in ((2304, 60000, 64), (1152, 38000, 32), (256, 60000, 64), (1024, 4000, 256), (256, 700, 256), (4608, 16000, 128)):
    a = torch.randn(k, m, generator=g).cuda(); b = torch.randn(k, n, generator=g).cuda()
    outs = [ops.mm(a.t(), b) for _ in range(6)]
    print('mm (%d x %d)^T (%d x %d): identical %s' % (k, m, k, n, all(torch.equal(o, outs[0]) for o in outs[1:])))
    outs = [torch.nn.functional.linear(b, a.t().contiguous()[:, :k]) if False else torch.mm(b.t(), a) for _ in range(6)]
    print('   torch.mm transposed form: identical %s' % all(torch.equal(o, outs[0]) for o in outs[1:]))
for (bsz, m, k, n) in ((24, 352, 352, 64), (24, 352, 64, 352), (6, 704, 256, 256)):
    a = torch.randn(bsz, m, k, generator=g).cuda(); b = torch.randn(bsz, k, n, generator=g).cuda()
    outs = [torch.bmm(a, b) for _ in range(6)]
    print('bmm %s: identical %s' % ((bsz, m, k, n), all(torch.equal(o, outs[0]) for o in outs[1:])))
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
torch.use_deterministic_algorithms(True, warn_only=True)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    out = model(dd, train=True, rng=np.random.default_rng(3))
    loss_fn(out, dd)['loss'].backward()
seen = sorted({str(x.message).split('.')[0][:160] for x in w if 'deterministic' in str(x.message)})
print('%d torch ops without a deterministic implementation:' % len(seen))
for s in seen: print('  ', s)
