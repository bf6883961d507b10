"""Host side of the training step (the step is bound by it: ~3 700 dispatches in ~50 ms): cProfile over five steps after three warm-up steps,
functions by own time.    python tools/r5/train_host_profile.py [rows]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd.data import registration_collate_fn_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
from se3et_amd.training import OverallLoss, make_optimizer, train_step

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45
dev = torch.device('cuda')
cfg = make_cfg('se3ete'); b = cfg.backbone
model = load_synthetic_weights(create_model(cfg)).to(dev).train()
loss_fn, opt = OverallLoss(cfg), make_optimizer(model, cfg, 1)
rng = np.random.RandomState(0)
pairs = []
for i in range(8):
    ref, src, T = make_pair('c2_5k', index=i)
    pairs.append(dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32), src_feats=np.ones((len(src), 1), np.float32), transform=T))


def step(i):
    dd = registration_collate_fn_stack_mode([pairs[i]], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits, device=dev)
    return train_step(model, dd, loss_fn, opt, rng=rng)


for i in range(3):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); c0 = time.process_time()
for i in range(3, 8):
    step(i)
torch.cuda.synchronize()
print('unprofiled: %.1f ms wall, %.1f ms process CPU per step' % ((time.perf_counter() - t0) * 200, (time.process_time() - c0) * 200))
pr = cProfile.Profile()
pr.enable()
for i in range(3, 8):
    step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(rows)
