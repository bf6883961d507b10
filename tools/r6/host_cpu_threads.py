"""Round 6: process CPU time of bench.py's three-batches-in-flight loop split by host thread (time.thread_time of every worker) against
time.process_time of the whole process: what the Python threads burn and what the runtime's own threads burn.
   python tools/r6/host_cpu_threads.py [--spin] [--no-ties] [--timing]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import se3et_amd
print('blocking waits:', se3et_amd.request_blocking_sync(0) if '--spin' not in sys.argv else 'not requested')
import numpy as np
import torch

from se3et_amd import _lib, ops
from se3et_amd.batched import forward_pairs
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair

if '--no-ties' in sys.argv:
    ops.RADIUS_REFERENCE_TIES = False
torch.cuda.set_device(0)
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg)).cuda().eval()
b = cfg.backbone
sys.setswitchinterval(1e-3)
batches = []
for s in range(12):
    clouds = []
    for j in range(8):
        ref, src, _ = make_pair('c2_5k', index=8 * s + j)
        clouds += [ref, src]
    batches.append((torch.from_numpy(np.concatenate(clouds, 0)).cuda(), torch.tensor([len(c) for c in clouds])))
ones = torch.ones((batches[0][0].shape[0], 1), device='cuda')
streams = [torch.cuda.Stream() for _ in range(3)]
spent = {}


def worker(t, steps):
    c0 = time.thread_time()
    with torch.cuda.stream(streams[t]), torch.no_grad():
        for i in steps:
            pts, lens = batches[i % len(batches)]
            d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
            d['features'] = ones
            forward_pairs(model, d)
        streams[t].synchronize()
    spent[t] = time.thread_time() - c0


def region(n):
    th = [threading.Thread(target=worker, args=(t, list(range(t, n, 3)))) for t in range(3)]
    c0, w0 = time.process_time(), time.perf_counter()
    [x.start() for x in th]
    [x.join() for x in th]
    torch.cuda.synchronize()
    return time.process_time() - c0, time.perf_counter() - w0


region(9)
if '--timing' in sys.argv:
    _lib.lib().se3_debug_kernel_timing(1)
for rep in range(2):
    cpu, wall = region(24)
    print('24 steps: wall %.1f ms per step, process CPU %.1f ms per step (%.2f cores); python worker threads %s ms per step; other threads %.1f ms per step' % (
        wall / 24 * 1e3, cpu / 24 * 1e3, cpu / wall, [round(v / 8 * 1e3, 1) for v in spent.values()], (cpu - sum(spent.values())) / 24 * 1e3), flush=True)
