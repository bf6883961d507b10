"""Sinkhorn (100 iterations, dustbins) on the patch pairs of a bench step: time per call and the largest difference between the scaling form
(score range within 40 of every row's maximum) and the log-domain loop (forced by one wide row per patch).  python tools/micro/sinkhorn_rate.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd import functional as SF
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for B, R in ((2048, 64), (1024, 64), (256, 64), (512, 128)):
    scores = (torch.randn(B, R, R, generator=g) * 3).cuda()
    rm, cm = (torch.rand(B, R, generator=g) < 0.8).cuda(), (torch.rand(B, R, generator=g) < 0.8).cuda()
    rm[:, 0] = True; cm[:, 0] = True
    alpha = torch.tensor(1.3).cuda()
    t = timeit(lambda: SF.log_optimal_transport(scores, rm, cm, alpha, 100, 1e12))
    wide = scores.clone(); wide[:, 0, 0] = 0.0; wide[:, 0, 1] = -50.0        # a valid row with a 50-wide range: the log-domain loop
    out_s = SF.log_optimal_transport(scores, rm, cm, alpha, 100, 1e12)
    base = scores.clone(); base[:, 0, 1] = -39.0
    narrow = SF.log_optimal_transport(base, rm, cm, alpha, 100, 1e12)
    base2 = base.clone(); base2[:, 0, 1] = -39.0; base2[:, 0, 2] = scores[:, 0, 2]
    tw = timeit(lambda: SF.log_optimal_transport(wide, rm, cm, alpha, 100, 1e12))
    print('%4d patch pairs of %3d x %3d: %7.1f us (scaling form)  %7.1f us (log-domain loop)' % (B, R, R, t, tw))
