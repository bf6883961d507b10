"""CPU probe: distinct neighbour rows of a 16-point tile when the query points of a KPConv table are taken in Morton order (per cloud) --
sizes the LDS row buffer of the union-staged KPConv producers.  Pyramid from the C oracle.  usage: union_sizes.py [c2_5k|demo] [tile]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import se3et_oracle as O
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair

which = sys.argv[1] if len(sys.argv) > 1 else 'c2_5k'
TP = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = make_cfg('se3ete'); b = cfg.backbone
if which == 'demo':
    ref = np.load('/root/reference/data/demo/ref.npy').astype(np.float32); src = np.load('/root/reference/data/demo/src.npy').astype(np.float32)
else:
    ref, src, _ = make_pair(which, 0)
pts = torch.from_numpy(np.concatenate([ref, src], 0)); lengths = torch.tensor([len(ref), len(src)])
dd = O.precompute(pts, lengths, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)

def spread(v):
    out = np.zeros_like(v)
    for i in range(16):
        out |= ((v >> i) & 1) << (3 * i)
    return out

def order(p, lens, cell, kind):
    q = np.floor(p / cell).astype(np.int64) + 32768
    if kind == 'morton':
        key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    else:
        return np.arange(len(p))
    cloud = np.repeat(np.arange(len(lens)), lens)
    return np.argsort(key | (cloud << 48), kind='stable')

S = b.num_stages
for name, qs, ss, tab in [('self', s, s, dd['neighbors'][s]) for s in range(S)] + [('strided', s + 1, s, dd['subsampling'][s]) for s in range(S - 1)]:
    q = dd['points'][qs].numpy(); lens = dd['lengths'][qs].numpy(); idx = tab.numpy(); Ns = len(dd['points'][ss])
    for kind, cellmul in (('hash', 1), ('morton', 1), ('morton', 2), ('morton', 4)):
        perm = order(q, lens, b.init_voxel_size * 2 ** qs * cellmul, kind)
        U = []; nv = []
        for t0 in range(0, len(perm), TP):
            rows = idx[perm[t0:t0 + TP]]
            v = rows[(rows >= 0) & (rows < Ns)]
            U.append(len(np.unique(v))); nv.append(len(v))
        U = np.array(U); nv = np.array(nv)
        print('%s %-8s q-stage %d s-stage %d P %6d NN %2d  %-7s cell x%d: entries/tile %5.0f  union mean %5.1f  p50 %3d p90 %3d p99 %3d max %3d  >96: %4.1f%%  >128: %4.1f%%  >160: %4.1f%%'
              % (which, name, qs, ss, len(q), idx.shape[1], kind, cellmul, nv.mean(), U.mean(), np.percentile(U, 50), np.percentile(U, 90), np.percentile(U, 99), U.max(),
                 100 * (U > 96).mean(), 100 * (U > 128).mean(), 100 * (U > 160).mean()))
