"""Round 6: what the reference-exact tie pass costs on the reference's real pair (tests/golden/demo_se3ete.npz: 18 977 + 15 953 points) and
that it costs nothing on a jittered synthetic pair.  Prints pyramid times (median of 7) with ops.RADIUS_REFERENCE_TIES on / off, the rows
flagged per table, and the forward's time."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3et_amd import ops
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'demo_se3ete.npz'))
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg)).cuda().eval()
b = cfg.backbone


def timed(f, n=7):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, r


for name, clouds in (('data/demo real pair', [g['ref'], g['src']]), ('synthetic c2_5k pair', list(make_pair('c2_5k', index=0)[:2]))):
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
    lens = torch.tensor([len(c) for c in clouds])
    limits = [38, 36, 36, 38]
    build = lambda: precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, limits)
    for flag in (True, False, True):
        ops.RADIUS_REFERENCE_TIES = flag
        build()
        ms, dd = timed(build)
        print('%-22s pyramid %7.2f ms  (RADIUS_REFERENCE_TIES = %s)' % (name, ms, flag))
    # rows flagged per search of stage 0 (the largest): one search with its own flags
    flags = torch.zeros(pts.shape[0] + 1, dtype=torch.int32, device='cuda')
    ops.radius_neighbors(pts, pts, lens, lens, b.init_radius, 38, ties=(flags[1:], flags[:1]))
    print('%-22s stage-0 rows flagged: %d of %d' % (name, int(flags[0]), pts.shape[0]))
    dd['features'] = torch.ones((pts.shape[0], 1), device='cuda')
    model(dd)
    ms, _ = timed(lambda: model(dd), 5)
    print('%-22s forward %7.2f ms' % (name, ms))

# ---- where the time of the tie pass goes (stage 0 of the real pair: 34 930 points, limit 38)
import ctypes
from se3et_amd._lib import lib
pts = torch.from_numpy(np.concatenate([g['ref'], g['src']], 0)).cuda()
lens = torch.tensor([len(g['ref']), len(g['src'])])
ops.RADIUS_REFERENCE_TIES = True
ms, host = timed(lambda: pts.cpu())
print('stage 0: device -> host copy of the points      %7.2f ms' % ms)
cap = lib().se3_kdtree_max_bytes(pts.shape[0], 2)
buf = torch.empty(cap, dtype=torch.uint8)
used = ctypes.c_size_t(0)
l64 = (ctypes.c_int64 * 2)(*lens.tolist())
ms, _ = timed(lambda: lib().se3_kdtree_build_host(host.data_ptr(), pts.shape[0], l64, 2, buf.data_ptr(), cap, ctypes.byref(used)))
print('stage 0: reference k-d tree on the host (1 thread) %7.2f ms, %d bytes' % (ms, used.value))
ms, tree = timed(lambda: ops.ReferenceTree(pts, lens))
print('stage 0: ops.ReferenceTree (copy + build + upload) %7.2f ms' % ms)
flags = torch.zeros(pts.shape[0] + 1, dtype=torch.int32, device='cuda')
full, mc = ops.radius_neighbors(pts, pts, lens, lens, b.init_radius, 38, ties=(flags[1:], flags[:1]))
n_tie, hits = int(flags[0]), int(mc.max())
ms, _ = timed(lambda: ops.radius_tie_order(full, pts, pts, lens, lens, b.init_radius, flags[1:], n_tie, hits, tree=tree))
print('stage 0: tie-order kernel, %d rows, max %d matches  %7.2f ms' % (n_tie, hits, ms))
ms, _ = timed(lambda: ops.radius_neighbors(pts, pts, lens, lens, b.init_radius, 38))
print('stage 0: the plain search                          %7.2f ms' % ms)
