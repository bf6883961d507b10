import os, sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
for name in ('neighbors', 'subsampling'):
    for i, idx in enumerate(dd[name]):
        ns = dd['points'][i].shape[0]
        v = ((idx >= 0) & (idx < ns)).sum(1).float()
        print(name, i, tuple(idx.shape), 'valid mean %.1f' % v.mean().item(), 'p50 %d' % v.median().item(), 'max %d' % v.max().item(), '>32: %.1f%%' % (100 * (v > 32).float().mean().item()), '>24: %.1f%%' % (100 * (v > 24).float().mean().item()), '<=16: %.1f%%' % (100 * (v <= 16).float().mean().item()), '<=8: %.1f%%' % (100 * (v <= 8).float().mean().item()))
