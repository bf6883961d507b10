import os, sys; sys.path.insert(0, '/root/repo')
import torch
from se3et_amd import ops
dev = torch.device('cuda')
def timeit(f, n=50):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for nseg in (8, 1):
  for rows in (310452, 100000, 30000, 9600, 3000):
    C = 64
    x = torch.randn(rows, C, device=dev); w = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    seg = ([rows * i // nseg // 6 * 6 for i in range(nseg)] + [rows]) if nseg > 1 else None
    t1 = timeit(lambda: ops.group_norm_stats(x, w, b, 32, 1e-5, segments=seg))
    t3 = timeit(lambda: ops.group_norm_rows(x, w, b, 32, 1e-5, 0.1, None, None, seg))
    print('segments %d rows %7d: stats %6.1f us   three-launch GroupNorm %6.1f us' % (nseg, rows, t1, t3))
