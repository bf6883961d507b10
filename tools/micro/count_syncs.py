"""Host synchronisations of one 8-pair forward and of one single-pair forward, by source line (torch sync-debug mode; not a test)."""
import sys, warnings, collections, traceback; sys.path.insert(0, '.')
import numpy as np, torch
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
from se3et_amd.batched import forward_pairs
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
def batch(k):
    clouds = []
    for j in range(8):
        ref, src, _ = make_pair('c2_5k', index=8 * k + j); clouds += [ref, src]
    return torch.from_numpy(np.concatenate(clouds, 0)).to(dev), torch.tensor([len(c) for c in clouds], dtype=torch.int64)
b = cfg.backbone
feats = None
def step(pts, lens):
    data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    data['features'] = torch.ones((pts.shape[0], 1), device=dev)
    return forward_pairs(model, data)
p0, p1 = batch(0), batch(1)
step(*p0); torch.cuda.synchronize()
counts = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    for fr in reversed(traceback.extract_stack()):
        if '/se3et_amd/' in fr.filename:
            counts['%s:%d %s' % (fr.filename.split('/se3et_amd/')[1], fr.lineno, fr.line.strip()[:90])] += 1
            return
    counts['%s:%d' % (filename, lineno)] += 1
warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step(*p1)
torch.cuda.set_sync_debug_mode('default')
for k, v in counts.most_common(): print('%3d  %s' % (v, k))
print('total', sum(counts.values()))
counts.clear()
ref, src, _ = make_pair('c2_5k', index=99)
pts1 = torch.from_numpy(np.concatenate([ref, src], 0)).to(dev); lens1 = torch.tensor([len(ref), len(src)])
def step1():
    data = precompute_data_stack_mode(pts1, lens1, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    data['features'] = torch.ones((pts1.shape[0], 1), device=dev)
    return model(data)
step1(); torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
step1()
torch.cuda.set_sync_debug_mode('default')
print('single pair:')
for k, v in counts.most_common(): print('%3d  %s' % (v, k))
print('total', sum(counts.values()))
