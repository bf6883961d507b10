"""Library GEMM time of one 8-pair bench step by call site shape: wraps torch.mm / addmm / baddbmm / bmm / matmul / einsum / F.linear with
HIP events (one forward, after warm-up) and prints the shapes sorted by total time.  python tools/gemm_shapes.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections, traceback
import numpy as np, torch, torch.nn.functional as F
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
log = []
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = orig(*a, **k); e1.record()
        shapes = tuple(tuple(t.shape) for t in a if torch.is_tensor(t))
        site = next((('%s:%d' % (os.path.basename(fr.filename), fr.lineno)) for fr in reversed(traceback.extract_stack()[:-1])
                     if 'se3et_amd' in fr.filename), '?')
        log.append((name, a[0] if isinstance(a[0], str) else '', shapes, site, e0, e1))
        return out
    setattr(mod, name, f)
for n in ('mm', 'addmm', 'baddbmm', 'bmm', 'matmul', 'einsum'): wrap(torch, n)
wrap(F, 'linear')
with torch.no_grad(): step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, eq, shapes, site, e0, e1 in log:
    k = (name, eq, shapes, site); v = agg.setdefault(k, [0, 0.0]); v[0] += 1; v[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
for (name, eq, shapes, site), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    fl = ''
    print('%7.3f ms  x%-3d %-8s %-22s %-28s %s' % (ms, n, name, eq, site, shapes))
print('total %.2f ms in %d calls (event brackets include launch gaps)' % (tot, len(log)))
