"""Which library GEMMs run in one 8-pair forward, with shapes and GPU time (torch.profiler; not a test)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
from se3et_amd.batched import forward_pairs
dev = torch.device('cuda')
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j)
    clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
lens = torch.tensor([len(c) for c in clouds], dtype=torch.int64)
b = cfg.backbone
def step():
    data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    data['features'] = torch.ones((pts.shape[0], 1), device=dev)
    return forward_pairs(model, data)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::linear', 'aten::matmul', 'aten::einsum'):
        dt = getattr(e, 'device_time_total', None)
        if dt is None: dt = e.cuda_time_total
        sdt = getattr(e, 'self_device_time_total', None)
        if sdt is None: sdt = e.self_cuda_time_total
        rows.append((sdt, e.count, e.key, str(e.input_shapes)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total self GPU time of GEMM ops: %.0f us per forward (8 pairs)' % tot)
for sdt, cnt, key, shp in rows[:45]:
    print('%8.0f us  x%-3d %-12s %s' % (sdt, cnt, key, shp))
