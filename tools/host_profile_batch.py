"""Host-side cost of one 8-pair step (pyramid + batched.forward_pairs) in ONE thread: cProfile over 10 steps, top functions by cumulative and
by own time (the GPU runs behind; the only waits are the step's own host synchronisations).   python tools/host_profile_batch.py [pairs]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from se3et_amd.batched import forward_pairs
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).cuda().eval(); b = cfg.backbone
clouds = []
for j in range(pairs):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda(); lens = torch.tensor([len(c) for c in clouds])
def step():
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = torch.ones((pts.shape[0], 1), device='cuda')
    return forward_pairs(model, d)
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print('wall %.2f ms per step, %.2f ms of process CPU time' % ((time.perf_counter() - t0) * 100, 0))
    c0 = time.process_time(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print('wall %.2f ms per step, %.2f ms of process CPU time per step' % ((time.perf_counter() - t0) * 100, (time.process_time() - c0) * 100))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): step()
    pr.disable(); torch.cuda.synchronize()
for key in ('cumulative', 'tottime'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28); print(s.getvalue()[:6000])
