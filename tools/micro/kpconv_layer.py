"""One KPConv layer of the bench shape (8 pairs per forward, real pyramid) on one path, a few calls: the workload for rocprofv3 --pmc passes.
python tools/micro/kpconv_layer.py <layer 0..9> [fused|union|sums|gemm] [calls]      (union: csrc/kpconv_union.hip, whatever the dispatch policy says)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd import ops, functional as SF, tables
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair
layer = int(sys.argv[1]); path = sys.argv[2] if len(sys.argv) > 2 else 'fused'; calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
ops.KPCONV_UNION = ops.KPCONV_UNION_ALL = True          # (orders for every stage while the pyramid is built; the path is picked below)
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
kidx = torch.from_numpy(tables.kernel_slot_table()).to(dev); ridx = torch.from_numpy(tables.anchor_slot_table()).to(dev)
calls_ = [(0, 0, 'neighbors', 32), (1, 0, 'subsampling', 32), (1, 1, 'neighbors', 64), (1, 1, 'neighbors', 64), (2, 1, 'subsampling', 64),
          (2, 2, 'neighbors', 128), (2, 2, 'neighbors', 128), (3, 2, 'subsampling', 128), (3, 3, 'neighbors', 256), (3, 3, 'neighbors', 256)]
qs, ss, tab, C = calls_[layer]
g = torch.Generator(device='cpu').manual_seed(0)
q, s = dd['points'][qs], dd['points'][ss]
idx = dd[tab][qs if tab == 'neighbors' else ss]
x = torch.randn(s.shape[0], 6, C, generator=g).to(dev)
w = (torch.randn(6, 6, C, C, generator=g) / (36 * C) ** 0.5).to(dev)
kp = torch.from_numpy(tables.kernel_points(b.init_radius * 2 ** ss)).to(dev)
sig = b.init_sigma * 2 ** ss
ops.KPCONV_MATRIX_CORE = {'fused': True, 'union': True, 'sums': 'sums', 'gemm': False}[path]
ops.KPCONV_UNION = path == 'union'
valid = ((idx >= 0) & (idx < s.shape[0])).sum(1).float()
with torch.no_grad():
    for _ in range(calls):
        y = SF.kpconv_inter_so3(x, q, s, idx, kp, w, kidx, ridx, sig)
torch.cuda.synchronize()
print('layer %d: P %d NN %d (valid: mean %.1f max %d) C %d path %s' % (layer, q.shape[0], idx.shape[1], float(valid.mean()), int(valid.max()), C, path))
