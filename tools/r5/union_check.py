"""GPU check + timing of the union-staged fused KPConv (csrc/kpconv_union.hip) against the per-lane-gather fused kernel (csrc/kpconv_mfma.hip)
and the slot-sum + library GEMM path: every KPConv layer of the 8-pair C2 pyramid (plain and kernel-specific layouts), one pair (channel split),
and random volumes with dense neighbourhoods (unions beyond 160 rows: sub-tiles).  python tools/r5/union_check.py [quick]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd import ops, functional as SF, tables
from se3et_amd._lib import lib, check
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
ops.KPCONV_UNION = ops.KPCONV_UNION_ALL = True          # every layer on the union-staged kernel, whatever the default / the dispatch policy say
kidx = torch.from_numpy(tables.kernel_slot_table()).to(dev); ridx = torch.from_numpy(tables.anchor_slot_table()).to(dev)
quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def plan_stats(q, G, NN):
    """sub-tiles per group and union rows, read back from the cached plan of the current stream"""
    hit = ops._union_plan_cache.get(ops._stream().value)
    plan = hit[2].cpu().numpy()
    a16 = lambda v: (v + 15) & ~15
    o = a16(G * 64)
    nsub = plan[o:o + 4 * G].view(np.int32)
    o += a16(4 * G)
    desc = plan[o:o + G * 16 * 16].view(np.int32).reshape(G, 16, 4)
    first = desc[np.arange(G), 0, 3]
    return 'groups %d  sub-tiles/group %.3f (max %d)  union of single-pass groups: mean %.1f max %d' % (
        G, nsub.mean(), nsub.max(), first[nsub == 1].mean() if (nsub == 1).any() else 0, first.max())


def compare(tag, x, q, s, idx, kp, w, sig, n=10):
    g = torch.Generator(device='cpu').manual_seed(1)
    P = q.shape[0]; C = x.shape[2]; Co = w.shape[-1]
    ref = torch.mm(ops.kpconv_slot_sums(x, q, s, idx, kp, kidx, ridx, sig), w.reshape(36 * C, Co)).view(P, 6, Co)
    res = {}
    for name, flag in (('fused', False), ('union', True)):
        ops.KPCONV_UNION = flag
        out = SF.kpconv_inter_so3(x, q, s, idx, kp, w, kidx, ridx, sig)
        out2 = SF.kpconv_inter_so3(x, q, s, idx, kp, w, kidx, ridx, sig)
        err = float((out - ref).abs().max() / ref.abs().max())
        same = bool(torch.equal(out, out2))
        t = timeit(lambda: SF.kpconv_inter_so3(x, q, s, idx, kp, w, kidx, ridx, sig), n)
        # the kernel-specific layout of x
        kind = 2 if flag else 1
        tb, errb = float('nan'), float('nan')
        if C % (8 if flag else 16) == 0:
            Ns = s.shape[0]
            if kind == 2:
                xb = x.view(Ns, 6, C // 8, 8).permute(0, 2, 1, 3).contiguous()
            else:
                xb = x.view(Ns, 3, 2, C // 16, 16).permute(0, 3, 1, 4, 2).contiguous()
            bf = ops.BlockedFeatures(xb, x.shape, kind)
            assert torch.equal(bf.plain(), x)
            outb = SF.kpconv_inter_so3(bf, q, s, idx, kp, w, kidx, ridx, sig)
            errb = float((outb - out).abs().max())
            tb = timeit(lambda: SF.kpconv_inter_so3(bf, q, s, idx, kp, w, kidx, ridx, sig), n)
        res[name] = (t, tb, err, errb, same, out)
    ops.KPCONV_UNION = True
    d = float((res['fused'][5] - res['union'][5]).abs().max() / ref.abs().max())
    po = ops.point_order(q)
    print('%-22s P %6d NN %2d C %3d->%3d  fused %.3f (blocked %.3f) ms err %.1e  | union %.3f (chunked %.3f) ms err %.1e  layout diff %.1e / %.1e  repeat-identical %s %s  union vs fused %.1e'
          % (tag, P, idx.shape[1], C, Co, res['fused'][0], res['fused'][1], res['fused'][2], res['union'][0], res['union'][1], res['union'][2],
             res['fused'][3], res['union'][3], res['fused'][4], res['union'][4], d), flush=True)
    if po is not None:
        print('    ' + plan_stats(q, po[1], idx.shape[1]), flush=True)
    return res['fused'][1] if res['fused'][1] == res['fused'][1] else res['fused'][0], res['union'][1] if res['union'][1] == res['union'][1] else res['union'][0]


def pyramid(nb):
    clouds = []
    for j in range(nb):
        ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
    return precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)


calls = [(0, 0, 'neighbors', 32), (1, 0, 'subsampling', 32), (1, 1, 'neighbors', 64), (2, 1, 'subsampling', 64),
         (2, 2, 'neighbors', 128), (3, 2, 'subsampling', 128), (3, 3, 'neighbors', 256)]
mult = [1, 1, 2, 1, 2, 1, 2]
g = torch.Generator(device='cpu').manual_seed(0)
for nb in ((1,) if quick else (1, 8)):
    dd = pyramid(nb)
    tf = tu = 0.0
    for (qs, ss, tab, C), m in zip(calls, mult):
        q, s = dd['points'][qs], dd['points'][ss]
        idx = dd[tab][qs if tab == 'neighbors' else ss]
        x = torch.randn(s.shape[0], 6, C, generator=g).to(dev)
        w = (torch.randn(6, 6, C, C, generator=g) / (36 * C) ** 0.5).to(dev)
        kp = torch.from_numpy(tables.kernel_points(b.init_radius * 2 ** ss)).to(dev)
        a, c = compare('%d pair(s) %s %d<-%d' % (nb, tab[:5], qs, ss), x, q, s, idx, kp, w, b.init_sigma * 2 ** ss)
        tf += m * a; tu += m * c
    print('== %d pair(s): all 10 layers  fused %.3f ms  union %.3f ms' % (nb, tf, tu), flush=True)

# random volumes: dense neighbourhoods, unions beyond the cap (sub-tiles), several clouds of odd sizes, padding rows, a cloud smaller than a tile
from se3et_amd import functional as SF2
for (sizes, radius, limit, C, Co) in (((1500, 37, 9, 700), 0.16, 64, 16, 32), ((900, 800), 0.22, 40, 32, 64), ((300,), 0.5, 33, 8, 96)):
    gen = np.random.default_rng(5)
    pts = torch.from_numpy(gen.uniform(0, 1, (sum(sizes), 3)).astype(np.float32)).to(dev)
    lens = torch.tensor(sizes)
    idx, _ = ops.radius_neighbors(pts, pts, lens, lens, radius, limit)
    idx = idx.contiguous()
    ops.register_point_order(pts, lens, radius / 2.5)
    x = torch.randn(pts.shape[0], 6, C, generator=g).to(dev)
    w = (torch.randn(6, 6, C, Co, generator=g) / (36 * C) ** 0.5).to(dev)
    kp = torch.from_numpy(tables.kernel_points(radius)).to(dev)
    compare('volume %s r %.2f' % (sizes, radius), x, pts, pts, idx, kp, w, radius / 2.5, n=3)
print('done')
