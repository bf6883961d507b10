import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from se3et_amd.data import registration_collate_fn_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
variant = sys.argv[1] if len(sys.argv) > 1 else 'se3ete'; pair = sys.argv[2] if len(sys.argv) > 2 else 'c2_5k'
cfg = make_cfg(variant); model = load_synthetic_weights(create_model(cfg)).cuda().eval()
ref, src, T = make_pair(pair)
d = dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32), src_feats=np.ones((len(src), 1), np.float32), transform=T)
def step():
    dd = registration_collate_fn_stack_mode([d], cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius, cfg.neighbor_limits)
    return dd, model(dd)
for _ in range(3): dd, out = step()
print([l.tolist() for l in dd['lengths']], [n.shape[1] for n in dd['neighbors']])
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): step()
torch.cuda.synchronize(); print('ms/pair', (time.time() - t0) * 100)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=28, max_name_column_width=50))
