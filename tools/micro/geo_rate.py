"""Geometric embedding at the bench shape: time per cloud and write rate.  python tools/micro/geo_rate.py [N]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd.model import create_model, make_cfg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 358
dev = torch.device('cuda'); torch.manual_seed(0)
cfg = make_cfg('se3ete'); model = create_model(cfg).to(dev).eval()
emb = model.transformer.embedding
pts = torch.rand(1, N, 3, device=dev) * 3.0
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    tabs = emb.tables()
    us = timeit(lambda: emb(pts, tabs))
    out = emb(pts, tabs)
mb = sum(o.numel() * 4 for o in (out if isinstance(out, (tuple, list)) else [out]) if o is not None) / 1e6
print('N %d  %.1f MB  %.1f us  %.2f TB/s written' % (N, mb, us, mb / us))
