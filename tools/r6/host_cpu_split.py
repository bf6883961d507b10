"""Round 6: process CPU time of one 8-pair step (one host thread, blocking waits requested first) split into the pyramid and the forward,
with the tie machinery on and off.  python tools/r6/host_cpu_split.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import se3et_amd
print('blocking waits:', se3et_amd.request_blocking_sync(0) if '--spin' not in sys.argv else 'not requested')
import numpy as np
import torch

from se3et_amd import ops
from se3et_amd.batched import forward_pairs
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair

cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg)).cuda().eval()
b = cfg.backbone
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j)
    clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
lens = torch.tensor([len(c) for c in clouds])
ones = torch.ones((pts.shape[0], 1), device='cuda')
for ties in (True, False, True):
    ops.RADIUS_REFERENCE_TIES = ties
    cpu = {'pyramid': 0.0, 'forward': 0.0}
    wall = {'pyramid': 0.0, 'forward': 0.0}
    with torch.no_grad():
        for it in range(13):
            c0, w0 = time.process_time(), time.perf_counter()
            d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
            c1, w1 = time.process_time(), time.perf_counter()
            d['features'] = ones
            forward_pairs(model, d)
            torch.cuda.synchronize()
            c2, w2 = time.process_time(), time.perf_counter()
            if it >= 3:
                cpu['pyramid'] += c1 - c0; cpu['forward'] += c2 - c1
                wall['pyramid'] += w1 - w0; wall['forward'] += w2 - w1
    print('ties %-5s  pyramid: cpu %.2f ms wall %.2f ms   forward: cpu %.2f ms wall %.2f ms' % (
        ties, cpu['pyramid'] * 100, wall['pyramid'] * 100, cpu['forward'] * 100, wall['forward'] * 100), flush=True)
