"""Host time of a one-pair forward by torch operator (torch.profiler, CPU side: self time of every aten op and of the HIP runtime calls),
and what is left for the interpreter: python tools/single_pair_host_ops.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from torch.profiler import ProfilerActivity, profile
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).cuda().eval(); b = cfg.backbone
inputs = []
for i in range(24):
    ref, src, _ = make_pair('c2_5k', index=1000 + i)
    inputs.append((torch.from_numpy(np.concatenate([ref, src])).cuda(), torch.tensor([len(ref), len(src)])))
feats = torch.ones((inputs[0][0].shape[0], 1), device='cuda')
def one(i):
    pts, lens = inputs[i]
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = feats
    return model(d)
for i in range(12): one(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(12, 20): one(i)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 8
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    for i in range(20, 24): one(i)
    torch.cuda.synchronize(); wall_prof = (time.perf_counter() - t0) / 4
cpu = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CPU]
tot = sum(e.self_cpu_time_total for e in cpu) / 4
print('wall %.2f ms per pair (%.2f under the profiler); self CPU time inside torch operators and HIP runtime calls %.2f ms per pair' % (wall * 1e3, wall_prof * 1e3, tot / 1e3))
for e in sorted(cpu, key=lambda e: -e.self_cpu_time_total)[:28]:
    print('%-44s x%5.1f  %7.1f us per pair  (%.1f us per call)' % (e.key[:44], e.count / 4, e.self_cpu_time_total / 4, e.self_cpu_time_total / max(e.count, 1)))
