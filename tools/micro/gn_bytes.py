"""Bytes moved by the GroupNorm launches of one 8-pair forward (not a test): prints totals to set against the kernel times."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from se3et_amd import ops as _ops, functional as SF
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
from se3et_amd.batched import forward_pairs
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev); lens = torch.tensor([len(c) for c in clouds], dtype=torch.int64)
b = cfg.backbone
data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
data['features'] = torch.ones((pts.shape[0], 1), device=dev)
rec = []
orig = _ops.group_norm_rows
def hook(x, weight, bias, groups, eps, leaky_slope, residual, x_bias=None, segments=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(x, weight, bias, groups, eps, leaky_slope, residual, x_bias, segments)
    e1.record()
    rec.append((x.numel() * 4, residual is not None, tuple(x.shape), e0, e1))
    return out
_ops.group_norm_rows = hook
if hasattr(SF, '_ops'): pass
forward_pairs(model, data); torch.cuda.synchronize(); rec.clear()
forward_pairs(model, data); torch.cuda.synchronize()
n = len(rec); xb = sum(r[0] for r in rec); rb = sum(r[0] for r in rec if r[1])
print('%d GroupNorm calls; x bytes %.2f GB; partial pass reads %.2f GB; apply pass moves %.2f GB (read x [+ residual], write out)' % (n, xb / 1e9, xb / 1e9, (2 * xb + rb) / 1e9))
for nb, res, shp, e0, e1 in sorted(rec, key=lambda r: -r[3].elapsed_time(r[4])):
    us = e0.elapsed_time(e1) * 1e3
    print('%-22s residual %d: %6.1f us for %7.1f MB moved -> %.2f TB/s' % (shp, res, us, (3 + res) * nb / 1e6, (3 + res) * nb / us / 1e6))
