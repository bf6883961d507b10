"""Round 6: thread scaling of the CPU baseline on this host -- the oracle's SE3ET-E forward on the 5k+5k pair at
8 / 16 / 32 / 64 / 128 torch threads (1 warm-up + 3 timed pairs each, median).  Justifies (or not) bench.py's default cap of 16 threads."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from oracle import se3et_oracle as O
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair

cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg))
state = {k: v.detach() for k, v in model.state_dict().items()}
oc = O.OracleConfig.from_model_cfg(cfg)
print('host: %s, %d cores (affinity %d)' % (bench.cpu_model_name(), os.cpu_count(), len(os.sched_getaffinity(0))))


def one(i):
    ref, src, _ = make_pair('c2_5k', index=i)
    pts = torch.from_numpy(np.concatenate([ref, src], 0))
    t0 = time.perf_counter()
    data = O.precompute(pts, torch.tensor([len(ref), len(src)]), oc.num_stages, oc.init_voxel_size, oc.init_radius, oc.neighbor_limits)
    t1 = time.perf_counter()
    data['features'] = torch.ones((pts.shape[0], 1))
    with torch.no_grad():
        O.forward(state, oc, data)
    return time.perf_counter() - t0, t1 - t0


print('threads   s/pair (median of 3)   pyramid s   pairs/s')
for threads in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    if threads > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(threads)
    one(0)
    runs = sorted(one(1 + i) for i in range(3))
    t, pyr = runs[1]
    print('%7d   %10.3f             %8.3f   %7.3f' % (threads, t, pyr, 1.0 / t), flush=True)
