"""One plain dense layer on the streaming kernel, a few launches, for counter passes: python tools/r5/one_linear.py <rows> <K> <N>"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3et_amd import ops
rows, K, N = (int(v) for v in sys.argv[1:4])
x = torch.randn(rows, K, device='cuda')
w = torch.randn(N, K, device='cuda') / 16
b = torch.randn(N, device='cuda')
for _ in range(6):
    y = ops.linear_stream(x, w, b)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    y = ops.linear_stream(x, w, b)
e.record()
torch.cuda.synchronize()
us = a.elapsed_time(e) / 20 * 1e3
print('layer rows %d K %d N %d: %.1f us, %.0f TFLOP/s f16 executed (3 products), %.2f TB/s of activations' % (rows, K, N, us, 6.0 * rows * K * N / us / 1e6, 4.0 * rows * (K + N) / us / 1e6))
