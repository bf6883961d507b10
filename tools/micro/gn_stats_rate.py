"""The statistics passes over a KPConv output at the bench shapes (8 pairs per forward: the 11 convolution outputs of the pyramid): time per
call of se3_group_norm_stats on the raw tensor and through one pending stage, and the read rate.  python tools/micro/gn_stats_rate.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd import ops
dev = torch.device('cuda')
def timeit(f, n=30):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(480000, 32, 2), (310452, 64, 3), (128466, 128, 3), (33036, 256, 3)]
tot = [0.0, 0.0]
for rows, C, count in shapes:
    x = torch.randn(rows, C, device=dev); w = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    seg = [rows * i // 8 // 6 * 6 for i in range(8)] + [rows]
    a1 = ops.group_norm_stats(x, w, b, 32, 1e-5, segments=seg)
    ref = torch.stack([torch.nn.functional.group_norm(x[seg[i]:seg[i + 1]].t()[None], 32, w, b, 1e-5)[0].t() for i in range(8)][:1])
    got = x[seg[0]:seg[1]] * a1[0, 0] + a1[0, 1]
    err = float((got - ref[0]).abs().max())
    t1 = timeit(lambda: ops.group_norm_stats(x, w, b, 32, 1e-5, segments=seg))
    pend = ops.Pending(x, [a1], [0.1], seg)
    t2 = timeit(lambda: ops.group_norm_stats(pend, w, b, 32, 1e-5))
    mb = rows * C * 4 / 1e6
    tot[0] += t1 * count; tot[1] += t2 * count
    print('rows %7d C %3d  %6.1f MB  raw %6.1f us (%.2f TB/s)  through a pending stage %6.1f us (%.2f TB/s)  x%d per step  err %.1e'
          % (rows, C, mb, t1, mb / t1, t2, mb / t2, count, err))
print('per 8-pair step: raw %.3f ms, pending %.3f ms' % (tot[0] / 1e3, tot[1] / 1e3))
