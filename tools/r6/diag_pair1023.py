"""Round-6 diagnostic: synthetic pair c2_5k index 1023 gives non-finite backbone features -- which block first?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from se3et_amd import ops
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair

dev = torch.device('cuda', 0)
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
b = cfg.backbone
idx = int(os.environ.get('PAIR', 1023))
ref, src, _ = make_pair('c2_5k', index=idx)
pts = torch.from_numpy(np.concatenate([ref, src], 0)).to(dev)
lens = torch.tensor([len(ref), len(src)], dtype=torch.int64)
data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
data['features'] = torch.ones((pts.shape[0], 1), dtype=torch.float32, device=dev)


def concrete(x):
    if isinstance(x, ops.Pending):
        return None                      # (statistics pending: checked at the consumer)
    if isinstance(x, ops.BlockedFeatures):
        return x.plain()
    return x if torch.is_tensor(x) else None


seen = []


def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, (tuple, list)) else out
        v = concrete(o)
        kind = type(o).__name__
        if isinstance(o, ops.Pending):
            r = o.raw
            fin = torch.isfinite(r)
            seen.append((name, 'Pending.raw', tuple(r.shape), int((~fin).reshape(r.shape[0], -1).any(1).sum()), float(r[fin].abs().max()) if fin.any() else None,
                         [bool(torch.isfinite(a).all()) for a in o.affines]))
            return
        if v is not None:
            fin = torch.isfinite(v)
            bad_rows = int((~fin).reshape(v.shape[0], -1).any(1).sum())
            seen.append((name, kind, tuple(v.shape), bad_rows, float(v[fin].abs().max()) if fin.any() else None))
        else:
            seen.append((name, kind, None, None, None))
    return f


for name, mod in model.backbone.named_modules():
    if name and (name.count('.') <= 1 or name.startswith('encoder4_1')):
        mod.register_forward_hook(hook(name))
ops.dense_saturated_rows(reset=True)
for packed in (False,):
    seen.clear()
    model.packed_inference = packed
    out = model(data, with_registration=False)
    torch.cuda.synchronize()
    print('packed_inference', packed, 'feats_f finite', bool(torch.isfinite(out['feats_f']).all()), 'feats_c finite', bool(torch.isfinite(out['feats_c']).all()),
          'dense_sat', ops.dense_saturated_rows())
    for s in seen:
        print('   ', s)

# ---- encoder4_1 by hand (its _forward_pending, piece by piece)
from se3et_amd import functional as SF
bb = model.backbone
blk = bb.encoder4_1
x_in = None


def grab(mod, inp, out):
    global x_in
    x_in = out


h = bb.encoder3_3.register_forward_hook(grab)
model.packed_inference = False
model(data, with_registration=False)
h.remove()
pts_l, sub = data['points'], data['subsampling']
q_pts, s_pts, nbi = pts_l[3], pts_l[2], sub[2]


def fin(name, t):
    t = t.plain() if isinstance(t, ops.BlockedFeatures) else t
    f = torch.isfinite(t)
    bad = (~f).reshape(t.shape[0], -1).any(1)
    print('  %-28s %-18s bad rows %d %s  max|finite| %s' % (name, tuple(t.shape), int(bad.sum()), bad.nonzero().flatten()[:8].tolist(),
                                                          float(t[f].abs().max()) if f.any() else None), flush=True)


print('encoder4_1 by hand: rows', q_pts.shape[0], 'support', s_pts.shape[0], 'table', tuple(nbi.shape), 'pad entries', int((nbi == s_pts.shape[0]).sum()),
      'rows with only padding', int((nbi == s_pts.shape[0]).all(1).sum()))
fin('input x', x_in)
with torch.no_grad(), SF.norm_segments([[0, 10000], [0, pts_l[1].shape[0]], [0, pts_l[2].shape[0]], [0, pts_l[3].shape[0]]]):
    SF.norm_segments.at_stage(3, support=2)
    conv = blk.interso3.conv
    p1 = blk.unary1.pending(x_in)
    fin('unary1.pending raw', p1.raw)
    print('     affines finite', [bool(torch.isfinite(a).all()) for a in p1.affines], [float(a.abs().max()) for a in p1.affines])
    xa = SF.norm_apply(p1, blocked=True, union=SF.kpconv_takes_union(q_pts, s_pts, conv.in_channels, conv.out_channels))
    fin('unary1 applied', xa)
    c = conv(q_pts, s_pts, nbi, xa)
    fin('conv', c)
    y = blk.interso3.norm.pending(c, 0.1)
    print('     norm1 affines', [bool(torch.isfinite(a).all()) for a in y.affines], [float(a.abs().max()) for a in y.affines])
    y = blk.norm.pending(y, 0.1)
    print('     norm2 affines', [bool(torch.isfinite(a).all()) for a in y.affines], [float(a.abs().max()) for a in y.affines])
    skip = SF.neighbor_max_pool(x_in, nbi)
    fin('max-pooled shortcut', skip)
    p2 = blk.unary2.pending(y)
    fin('unary2.pending raw', p2.raw)
    print('     unary2 affines', [bool(torch.isfinite(a).all()) for a in p2.affines], [float(a.abs().max()) for a in p2.affines])
    out = SF.norm_apply(p2, residual=skip, final_slope=0.1)
    fin('block output', out)
    # ---- which elements, and what does the kernel see in that row
    raw = p2.raw.reshape(-1, 512)
    badr = (~torch.isfinite(raw)).any(1).nonzero().flatten().tolist()
    print('bad flat rows', badr, 'bad cols per row', [(int((~torch.isfinite(raw[r])).sum())) for r in badr])
    yin = y.raw.reshape(-1, 128)
    a1, a2 = y.affines
    t = yin * a1[0, 0] + a1[0, 1]
    t = torch.where(t > 0, t, t * y.slopes[0])
    t = t * a2[0, 0] + a2[0, 1]
    t = torch.where(t > 0, t, t * y.slopes[1])
    for r in badr[:3]:
        row = t[r]
        print('row', r, 'transformed: min|.|', float(row.abs().min()), 'max|.|', float(row.abs().max()), 'per 32-block max', [float(row[k:k + 32].abs().max()) for k in range(0, 128, 32)],
              'raw max', float(yin[r].abs().max()), 'slopes', y.slopes)
    print('rows whose first 32 transformed values are all below 2^-4:', int((t[:, :32].abs().amax(1) < 0.0625).sum()), (t[:, :32].abs().amax(1) < 0.0625).nonzero().flatten()[:10].tolist())
    print('rows with a value >= 128 in the first 32 / >= 32768 anywhere:', int((t[:, :32].abs().amax(1) >= 128).sum()), int((t.abs().amax(1) >= 32768).sum()))
    torch.save({'yin': yin.cpu(), 'a1': a1.cpu(), 'a2': a2.cpu(), 'w': blk.unary2.mlp.weight.detach().cpu()}, os.path.join(ROOT, 'gpurun_out', 'r6k', 'unary2_case.pt'))
    # ---- variations of the unary2 input: which row pays for the tiny row?
    def run(raw_mod, tag):
        P = ops.Pending(raw_mod.view(687, 6, 128).contiguous(), y.affines, y.slopes, y.segments)
        r = blk.unary2.pending(P).raw.reshape(-1, 512)
        bad = (~torch.isfinite(r)).any(1).nonzero().flatten().tolist()
        ref = None
        print('   %-46s bad rows %s' % (tag, bad), flush=True)
        return r
    base = run(yin.clone(), 'as is')
    m = yin.clone(); m[3283] = yin[3282]
    fixed = run(m, 'tiny row 3283 replaced by its neighbour')
    m = yin.clone(); m[3283], m[3315] = yin[3315], yin[3283]
    run(m, 'tiny row moved to 3315 (next 32-row block)')
    m = yin.clone(); m[3283], m[100] = yin[100], yin[3283]
    run(m, 'tiny row moved to 100')
    m = yin.clone(); m[3283], m[3284] = yin[3284], yin[3283]
    run(m, 'tiny row moved to 3284')
    # accuracy of the rows of that block against float64 when nothing is non-finite
    w = blk.unary2.mlp.weight.detach().double()
    want = t.double() @ w.t()
    err = (fixed.double() - torch.cat((want[:3283], (torch.where((yin[3282] * a1[0, 0] + a1[0, 1]) > 0, yin[3282] * a1[0, 0] + a1[0, 1], (yin[3282] * a1[0, 0] + a1[0, 1]) * 0.1)[None]).double() @ w.t() * 0, want[3284:]))).abs()
    err[3283] = 0
    print('   max error of the repaired run against float64 (bias-free product):', float(err.max()), 'at row', int(err.amax(1).argmax()), 'ref max', float(want.abs().max()))
