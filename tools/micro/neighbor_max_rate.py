"""The neighbour max-pool of the strided blocks on the tables of the real pyramid (8 pairs of the 5k preset, 4 of the KITTI one): time per
call.  python tools/micro/neighbor_max_rate.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd import ops
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair
def timeit(f, n=30):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for variant, preset, pairs, widths in (('se3ete', 'c2_5k', 8, (32, 128, 256)), ('se3eti_kitti', 'c3_20k', 4, (32, 128, 256, 512))):
    cfg = make_cfg(variant); b = cfg.backbone
    clouds = []
    for j in range(pairs):
        ref, src, _ = make_pair(preset, index=j); clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda(); lens = torch.tensor([len(c) for c in clouds])
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    tot = 0.0
    for i, t in enumerate(d['subsampling']):
        C = widths[i]
        x = torch.randn(d['points'][i].shape[0], 6, C, device='cuda')
        out = ops.neighbor_max_pool(x, t)
        n = x.shape[0]
        xp = torch.cat((x, torch.zeros_like(x[:1])), 0)
        tt = t.clamp(min=0)
        ref = xp[tt.clamp(max=n)].masked_fill((t < 0)[:, :, None, None], float('-inf')).amax(1)
        us = timeit(lambda: ops.neighbor_max_pool(x, t))
        tot += us
        print('%s stage %d: %d x %d neighbours, 6 x %d channels: %7.1f us  equal to the gather + amax: %s' % (variant, i, t.shape[0], t.shape[1], C, us, bool(torch.equal(out, ref))))
    print('%s: %.3f ms per step' % (variant, tot / 1e3))
