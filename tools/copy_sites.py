"""GPU time of torch copies / cats / fills in one 8-pair forward by se3et_amd call site (Python-level wrappers + HIP events).
python tools/copy_sites.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections, traceback
import numpy as np, torch
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
def site():
    st = [f for f in traceback.extract_stack()[:-2] if 'se3et_amd' in f.filename]
    return ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-2:]))
def wrap(owner, name, pred=None):
    orig = getattr(owner, name)
    def f(*a, **k):
        if pred is not None and not pred(*a, **k):
            return orig(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = orig(*a, **k); e1.record()
        shp = tuple(a[0].shape) if a and torch.is_tensor(a[0]) else (tuple(tuple(t.shape) for t in a[0]) if a and isinstance(a[0], (list, tuple)) and a[0] and torch.is_tensor(a[0][0]) else a[:1])
        log.append((name, site(), str(shp)[:60], e0, e1))
        return out
    setattr(owner, name, f)
wrap(torch.Tensor, 'contiguous', lambda t, *a, **k: t.is_cuda and not t.is_contiguous())
wrap(torch.Tensor, 'clone', lambda t, *a, **k: t.is_cuda)
wrap(torch.Tensor, 'to', lambda t, *a, **k: t.is_cuda)
wrap(torch.Tensor, 'float', lambda t, *a, **k: t.is_cuda and t.dtype != torch.float32)
wrap(torch.Tensor, 'reshape', lambda t, *a, **k: t.is_cuda and not t.is_contiguous())
for n in ('cat', 'stack', 'zeros', 'zeros_like', 'full', 'ones', 'empty_like'): wrap(torch, n)
wrap(torch.Tensor, 'new_zeros'); wrap(torch.Tensor, 'new_full'); wrap(torch.Tensor, 'masked_fill'); wrap(torch.Tensor, 'fill_'); wrap(torch.Tensor, 'zero_'); wrap(torch.Tensor, 'copy_')
with torch.no_grad(): step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, st, shp, e0, e1 in log:
    v = agg.setdefault((name, st, shp), [0, 0.0]); v[0] += 1; v[1] += e0.elapsed_time(e1) * 1e3
for (name, st, shp), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%8.1f us x%-3d %-12s %-50s %s' % (us, n, name, st, shp))
print('total %.2f ms (event brackets: includes launch gaps)' % (sum(v[1] for v in agg.values()) / 1e3))
