"""One pair per forward through SE3ET.forward (module path) and through batched.forward_pairs with B = 1 (packed-row kernels):
wall time per pair, kernel launches and summed kernel time.  python tools/single_pair_paths.py [pairs] [module | 'packed B=1']"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from se3et_amd import batched
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).cuda().eval(); b = cfg.backbone
inputs = []
for i in range(3 * (n + 5)):          # every path gets pairs of its own: the shapes of a timed forward have never been seen before
    ref, src, _ = make_pair('c2_5k', index=1000 + i)
    inputs.append((torch.from_numpy(np.concatenate([ref, src])).cuda(), torch.tensor([len(ref), len(src)])))
feats = torch.ones((inputs[0][0].shape[0], 1), device='cuda')
def data(i):
    pts, lens = inputs[i]
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = feats
    return d
def module_path(i):
    model.packed_inference = False
    try: return model(data(i))
    finally: model.packed_inference = True
def small_gemms_own(i):
    from se3et_amd import ops
    keep = ops.LINEAR_F16_MIN_ROWS; ops.LINEAR_F16_MIN_ROWS = 0
    try: return model(data(i))
    finally: ops.LINEAR_F16_MIN_ROWS = keep
paths = {'module': module_path, 'packed B=1': lambda i: batched.forward_pairs(model, data(i))[0], 'packed / own GEMM kernel for all row counts': small_gemms_own}
if len(sys.argv) > 2: paths = {k: paths[k] for k in sys.argv[2].split('|')}      # one path only (under rocprofv3)
outs = {}
for k, (name, f) in enumerate(paths.items()):
    base = k * (n + 5)
    for i in range(3): f(base + i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3, n + 3): o = f(base + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3, n + 3): o = f(base + i)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / n
    outs[name] = f(3)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        f(base + n + 3); torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ker = [e for e in ev if 'Memcpy' not in e.name and 'Memset' not in e.name]
    print('%-44s %.3f ms/pair (%.1f pairs/s; the same pairs again: %.1f)  kernels %d (+ %d copies/fills)  kernel time %.3f ms' % (
        name, dt * 1e3, 1 / dt, 1 / dt2, len(ker), len(ev) - len(ker), sum(e.device_time_total for e in ker) / 1e3))
if len(outs) < 2: sys.exit(0)
a, c = outs['module'], outs['packed B=1']
for k in ('estimated_transform', 'ref_feats_c', 'matching_scores', 'corr_scores'):
    print(k, tuple(a[k].shape), tuple(c[k].shape), float((a[k] - c[k]).abs().max()) if a[k].shape == c[k].shape else 'shape differs')
