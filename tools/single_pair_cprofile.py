"""Host time of one single-pair forward by Python function (cProfile, no synchronisation inside the forward):
python tools/single_pair_cprofile.py [pairs]"""
import cProfile, pstats, sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).cuda().eval(); b = cfg.backbone
inputs = []
for i in range(n + 3):
    ref, src, _ = make_pair('c2_5k', index=1000 + i)
    inputs.append((torch.from_numpy(np.concatenate([ref, src])).cuda(), torch.tensor([len(ref), len(src)])))
feats = torch.ones((inputs[0][0].shape[0], 1), device='cuda')
def one(i):
    pts, lens = inputs[i]
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = feats
    with torch.no_grad(): return model(d)
for i in range(3): one(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3, n + 3): one(i)
torch.cuda.synchronize(); print('wall ms/pair %.3f' % ((time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(3, n + 3): one(i)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)
