"""Where does a step of the union-staged KPConv go?  Times the kernel with parts switched off (se3_debug_set_kpconv_union_variant; results
are wrong with any bit set).  python tools/r5/union_variants.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd import ops, functional as SF, tables
from se3et_amd._lib import lib
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
ops.KPCONV_UNION = ops.KPCONV_UNION_ALL = True          # every layer on the union-staged kernel, whatever the default / the dispatch policy say
kidx = torch.from_numpy(tables.kernel_slot_table()).to(dev); ridx = torch.from_numpy(tables.anchor_slot_table()).to(dev)
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
calls = [(0, 0, 'neighbors', 32), (1, 0, 'subsampling', 32), (1, 1, 'neighbors', 64), (2, 1, 'subsampling', 64),
         (2, 2, 'neighbors', 128), (3, 2, 'subsampling', 128), (3, 3, 'neighbors', 256)]
g = torch.Generator(device='cpu').manual_seed(0)
names = {23: 'cons, weights in L1', 39: 'cons, no A reads', 55: 'cons, neither', 16: 'all, weights in L1', 0: 'all', 1: '-gather', 2: '-loads', 4: '-abuild', 8: '-consumer MFMA', 7: 'producers idle', 15: 'barriers only', 3: '-gather -loads', 9: '-gather -consumer'}
for qs, ss, tab, C in calls:
    q, s = dd['points'][qs], dd['points'][ss]
    idx = dd[tab][qs if tab == 'neighbors' else ss]
    Ns = s.shape[0]
    x = torch.randn(Ns, 6, C, generator=g).to(dev)
    bf = ops.BlockedFeatures(x.view(Ns, 6, C // 8, 8).permute(0, 2, 1, 3).contiguous(), x.shape, 2)
    w = (torch.randn(6, 6, C, C, generator=g) / (36 * C) ** 0.5).to(dev)
    kp = torch.from_numpy(tables.kernel_points(b.init_radius * 2 ** ss)).to(dev)
    sig = b.init_sigma * 2 ** ss
    f = lambda: SF.kpconv_inter_so3(bf, q, s, idx, kp, w, kidx, ridx, sig)
    ops.KPCONV_UNION = False
    bf1 = ops.BlockedFeatures(x.view(Ns, 3, 2, C // 16, 16).permute(0, 3, 1, 4, 2).contiguous(), x.shape, 1) if C % 16 == 0 else x
    told = timeit(lambda: SF.kpconv_inter_so3(bf1, q, s, idx, kp, w, kidx, ridx, sig))
    ops.KPCONV_UNION = True
    out = ['fused %.3f' % told]
    for v in (0, 1, 2, 4, 8, 7, 15):
        lib().se3_debug_set_kpconv_union_variant(v)
        out.append('%s %.3f' % (names[v], timeit(f)))
    lib().se3_debug_set_kpconv_union_variant(0)
    print('P %6d C %3d %s: ' % (q.shape[0], C, tab[:5]) + ' | '.join(out), flush=True)
