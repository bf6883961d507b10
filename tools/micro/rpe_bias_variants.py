"""RPE logits kernel at the bench shape (16 clouds per launch): f16-split MFMAs (default) against the exact f32 MFMAs (variant 3), with the
error of both against a float64 evaluation.  python tools/micro/rpe_bias_variants.py"""
import os, sys; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
import time, torch
import bench_attention_stack as B
from se3et_amd._lib import lib
x = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < 1.5: y = x @ x
torch.cuda.synchronize()
lengths = (382, 350, 304, 310, 382, 350, 304, 310, 382, 350, 304, 310, 382, 350, 304, 310)
for A, eq in ((6, True), (1, False)):
    for bv, av in ((0, 0), (0, 5), (0, 6), (0, 7), (0, 8), (0, 10)):
        B.run(A, lengths[:2], eq, bv, 0, av, iters=1, check=True)
    for rep in range(2):
        for bv, av in ((0, 0), (0, 5), (0, 6), (0, 7), (0, 8), (0, 10)):
            tb, ta, nbytes = B.run(A, lengths, eq, bv, 0, av, iters=20)
            print('A=%d eq=%d attention variant %2d: logits kernel %.1f us, attention kernel %.1f us' % (A, eq, av, tb, ta))
lib().se3_debug_set_bias_variant(0, 0); lib().se3_debug_set_attention_variant(0)
