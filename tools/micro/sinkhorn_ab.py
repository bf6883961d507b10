"""The two forms of the Sinkhorn forward kernel (csrc/sinkhorn.hip: base 2 with carried shifts / the reference's order of operations) on the
bench shape -- 2048 patch pairs of 64 x 64 points, 100 iterations -- and the KITTI one; time per call and the largest difference on valid
entries.    python tools/micro/sinkhorn_ab.py [one]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd import functional as SF
from se3et_amd._lib import lib


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator().manual_seed(0)
shapes = ((2048, 64, 64, 0.8, 3.0), (2048, 64, 64, 1.0, 1.0), (1024, 64, 64, 0.6, 12.0), (256, 128, 128, 0.7, 3.0))
one = len(sys.argv) > 1 and sys.argv[1] == 'one'          # counter passes (tools/pmc_kernel.sh): the bench shape only, three calls per form
for B, R, C, frac, scale in shapes[:1] if one else shapes:
    scores = (torch.randn(B, R, C, generator=g) * scale).cuda()
    rm, cm = (torch.rand(B, R, generator=g) < frac).cuda(), (torch.rand(B, C, generator=g) < frac).cuda()
    alpha = torch.tensor(1.0).cuda()
    outs, ms = [], []
    for variant in (1, 0):
        lib().se3_debug_set_sinkhorn_variant(variant)
        f = lambda: SF.log_optimal_transport(scores, rm, cm, alpha, 100, 1e12)
        outs.append(f()); ms.append(timeit(f, 2 if one else 10))
    lib().se3_debug_set_sinkhorn_variant(0)
    valid = outs[0] > -1e11
    assert torch.equal(outs[1] > -1e11, valid)
    d = float((outs[0][valid] - outs[1][valid]).abs().max() / outs[0][valid].abs().max())
    print('B %4d  %3d x %3d  valid %.1f  score scale %4.1f: reference order %.3f ms, base 2 + carried shifts %.3f ms  (x%.2f), difference %.1e' %
          (B, R, C, frac, scale, ms[0], ms[1], ms[0] / ms[1], d))
