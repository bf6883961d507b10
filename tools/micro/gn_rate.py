"""GroupNorm at the bench shapes (8 pairs per forward): time per call with the per-kernel split, bytes moved (2 reads + 1 write) per
second.  python tools/micro/gn_rate.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd import ops
dev = torch.device('cuda')
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
P = [80000, 51742, 21411, 7690]
shapes = [(0, 32), (0, 64), (0, 16), (1, 32), (1, 64), (1, 128), (2, 64), (2, 128), (2, 256), (3, 128), (3, 256), (3, 512)]
for div, nseg in ((1, 8), (8, 1)):          # 8 pairs per forward (8 segments), one pair per forward (one segment)
    for st, C in shapes:
        rows = P[st] * 6 // div
        x = torch.randn(rows, C, device=dev); w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        seg = [rows * i // nseg for i in range(nseg + 1)] if nseg > 1 else None
        us = timeit(lambda: ops.group_norm_rows(x, w, b, 32 if C >= 32 else C, 1e-5, 0.1, None, None, seg))
        mb = rows * C * 4 / 1e6
        print('segments %d rows %7d C %3d  %6.1f MB  %7.1f us  %.2f TB/s (3 passes)' % (nseg, rows, C, mb, us, 3 * mb / us))
