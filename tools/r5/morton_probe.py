"""Probe (GPU): how much do the backbone's gather kernels gain from spatially ordered rows?  Builds the 8-pair C2 pyramid, permutes every
stage's rows by a Morton key inside each cloud (pure torch, untimed), remaps the ten tables, and times model.backbone in both orders.
usage: morton_probe.py [morton|plain|both]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3et_amd import functional as SF
from se3et_amd.batched import _offsets
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair

mode = sys.argv[1] if len(sys.argv) > 1 else 'both'
preset, variant, B = (sys.argv[2], sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else ('c2_5k', 'se3ete', 8)
cfg = make_cfg(variant)
model = load_synthetic_weights(create_model(cfg), 7).cuda().eval()
clouds = []
for p in range(B):
    r, s, _ = make_pair(preset, index=p)
    clouds += [r, s]
pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
b = cfg.backbone
dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
dd['features'] = torch.ones((pts.shape[0], 1), device='cuda')


def spread(v):
    v = v & 0xFFFF
    out = torch.zeros_like(v)
    for i in range(16):
        out |= ((v >> i) & 1) << (3 * i)
    return out


def morton(dd, cell0):
    S = len(dd['points'])
    perms, invs = [], []
    for s in range(S):
        p = dd['points'][s]
        cell = cell0 * 2 ** s
        q = torch.floor(p / cell).long() + 32768
        key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        cloud = torch.repeat_interleave(torch.arange(len(dd['lengths'][s]), device=p.device), dd['lengths'][s].to(p.device))
        key = key | (cloud << 48)
        perm = torch.sort(key, stable=True)[1]
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(len(perm), device=p.device)
        perms.append(perm); invs.append(inv)
    out = {'lengths': dd['lengths'], 'points': [dd['points'][s][perms[s]].contiguous() for s in range(S)], 'neighbors': [], 'subsampling': [], 'upsampling': []}

    def remap(t, rows_perm, inv, n_support):
        t = t[rows_perm]
        pad = torch.tensor([n_support, -1], device=t.device)
        ext = torch.cat((inv, pad))                      # index n_support -> n_support, -1 -> -1
        return ext[t].contiguous()
    for s in range(S):
        out['neighbors'].append(remap(dd['neighbors'][s], perms[s], invs[s], len(perms[s])))
    for s in range(S - 1):
        out['subsampling'].append(remap(dd['subsampling'][s], perms[s + 1], invs[s], len(perms[s])))
        out['upsampling'].append(remap(dd['upsampling'][s], perms[s], invs[s + 1], len(perms[s + 1])))
    out['features'] = dd['features'][perms[0]].contiguous()
    return out, perms, invs


seg = []
for ln in dd['lengths']:
    o = _offsets(ln.tolist())
    seg.append([o[2 * p] for p in range(B)] + [o[-1]])


def run(d, n):
    with torch.no_grad(), SF.norm_segments(seg):
        for _ in range(3):
            f = model.backbone(d['features'], d)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(n):
            f = model.backbone(d['features'], d)
        torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3, f


if mode in ('plain', 'both'):
    ms, f0 = run(dd, 20)
    print('plain order : backbone %.3f ms' % ms)
if mode in ('morton', 'both', 'mortonx'):
    md, perms, invs = morton(dd, b.init_voxel_size)
    ms, f1 = run(md, 20)
    print('morton order: backbone %.3f ms' % ms)
    from se3et_amd._lib import lib
    lib().se3_debug_set_kpconv_variant(1)
    ms, f1 = run(md, 20)
    print('morton order, XCD-contiguous tiles: backbone %.3f ms' % ms)
    if mode == 'both':
        lib().se3_debug_set_kpconv_variant(0)
    if mode == 'both':
        c0, c1 = f0[-1], f1[-1][invs[-1]]
        print('feats_c agreement', float((c0 - c1).abs().max() / c0.abs().max()), 'feats_f', float((f0[0] - f1[0][invs[1]]).abs().max() / f0[0].abs().max()))
