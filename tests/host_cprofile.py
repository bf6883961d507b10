"""cProfile of the host side of one pair (not a test): python tests/host_cprofile.py"""
import sys; sys.path.insert(0, '.')
import cProfile, pstats, io
import numpy as np, torch
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).cuda().eval()
ref, src, T = make_pair('c2_5k')
pts = torch.from_numpy(np.concatenate([ref, src])).cuda(); lens = torch.tensor([len(ref), len(src)])
feats = torch.ones((pts.shape[0], 1), device='cuda')
b = cfg.backbone
def step():
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = feats
    return model(d)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize()
pr.disable()
for key in ('tottime', 'cumtime'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45 if key == 'tottime' else 60); print(s.getvalue()[:9000])
