"""Host-side cost per component (no sync inside): python tools/host_profile.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, collections
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
import se3et_amd.model as M
cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).cuda().eval()
ref, src, T = make_pair('c2_5k')
pts = torch.from_numpy(np.concatenate([ref, src])).cuda(); lens = torch.tensor([len(ref), len(src)])
feats = torch.ones((pts.shape[0], 1), device='cuda')
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(mod, name):
    orig = mod.forward
    def f(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
    mod.forward = f
wrap(model.backbone, 'backbone'); wrap(model.transformer, 'transformer(all)'); wrap(model.transformer.embedding, ' embedding')
wrap(model.coarse_matching, 'coarse_matching'); wrap(model.optimal_transport, 'sinkhorn'); wrap(model.fine_matching, 'lgr')
for i, l in enumerate(model.transformer.transformer.layers): wrap(l, ' layer(%s)' % cfg.geotransformer.blocks[i])
b = cfg.backbone
def step():
    t0 = time.perf_counter()
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    acc['precompute'] += time.perf_counter() - t0
    d['features'] = feats
    t0 = time.perf_counter(); out = model(d); acc['model(total)'] += time.perf_counter() - t0
for _ in range(3): step()
acc.clear(); cnt.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print('wall ms/pair', tot * 100)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print('%-28s %7.2f ms/pair  (%d calls)' % (k, v * 100, cnt[k] // 10))
