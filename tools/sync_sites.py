"""Host synchronisations of one single-pair forward (pyramid + model), by call site: python tools/sync_sites.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections, traceback, warnings
import numpy as np, torch
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
ref, src, _ = make_pair('c2_5k', index=0)
pts = torch.from_numpy(np.concatenate([ref, src], 0)).to(dev); lens = torch.tensor([len(ref), len(src)])
feats = torch.ones((pts.shape[0], 1), device=dev)
def step():
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = feats
    return model(d)
with torch.no_grad():
    for _ in range(2): step()
torch.cuda.synchronize()
sites = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' in str(message) and 'prototype' not in str(message):
        st = [f for f in traceback.extract_stack() if 'se3et_amd' in f.filename]
        if not st:
            st = traceback.extract_stack()[:-1]          # no library frame: show the caller's own
        sites[' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-3:]))] += 1
warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
with torch.no_grad(): step()
torch.cuda.set_sync_debug_mode('default')
for k, v in sites.items(): print(v, k)
