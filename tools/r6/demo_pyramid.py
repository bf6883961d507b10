"""Round 6: the pyramid of the real pair (tests/golden/demo_se3ete.npz) N times, for profilers (python3 tools/r6/demo_pyramid.py [N])."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3et_amd.data import precompute_data_stack_mode

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'demo_se3ete.npz'))
pts = torch.from_numpy(np.concatenate([g['ref'], g['src']], 0)).cuda()
lens = torch.tensor([len(g['ref']), len(g['src'])])
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    precompute_data_stack_mode(pts, lens, 4, 0.025, 0.0625, [38, 36, 36, 38])
    torch.cuda.synchronize()
    print('pyramid %d: %.2f ms' % (i, (time.perf_counter() - t0) * 1e3), flush=True)
