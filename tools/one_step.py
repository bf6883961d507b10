"""ONE bench step (8 stacked c2_5k pairs: on-GPU pyramid + forward_pairs incl. LGR) and nothing else -- the workload of the whole-step PMC
passes (tools/pmc_step.sh), where every dispatch costs ~0.1 s of counter collection.  The model's one-off work (weight pieces, embedding
tables: ~0.3 GB of traffic) is part of the process and of the counters; it is below 2 % of a step's traffic.
python tools/one_step.py [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3et_amd.batched import forward_pairs
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg)).cuda().eval()
b = cfg.backbone
for s in range(steps):
    clouds = []
    for j in range(8):
        ref, src, _ = make_pair('c2_5k', index=8 * s + j)
        clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
    data = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    data['features'] = torch.ones((pts.shape[0], 1), device='cuda')
    outs = forward_pairs(model, data)
torch.cuda.synchronize()
print('one_step: %d step(s), transform of pair 0:' % steps, outs[0]['estimated_transform'][0].tolist())
