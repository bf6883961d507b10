"""Which call sites of the 8-pair forward spend GPU time in torch's own small kernels (copies, cats, fills, element-wise)?  Wraps the
dispatcher with torch.profiler (with_stack) and groups self-CUDA time of aten ops by the innermost se3et_amd frame.
python tools/torch_ops_sites.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from se3et_amd.batched import forward_pairs
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev); lens = torch.tensor([len(c) for c in clouds])
feats = torch.ones((pts.shape[0], 1), device=dev)
def step():
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = feats
    return forward_pairs(model, d)
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    t = getattr(e, 'self_device_time_total', None)
    if t is None: t = getattr(e, 'self_cuda_time_total', 0)
    if not t or not e.name.startswith('aten::') or any(k in e.name for k in ('mm', 'linear', 'matmul', 'einsum', 'bmm')): continue
    site = next((s for s in (e.stack or []) if 'se3et_amd' in s), '?')
    site = site.split('se3et_amd/')[-1][:70]
    k = (e.name, site); agg[k][0] += 1; agg[k][1] += t
tot = sum(v[1] for v in agg.values())
for (name, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%8.1f us x%-3d %-28s %s' % (us, n, name, site))
print('total %.2f ms' % (tot / 1e3))
