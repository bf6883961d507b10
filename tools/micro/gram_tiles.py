"""gram_stack by pair count (A = 6, C = 256): python tools/micro/gram_tiles.py
se3_gram_stack takes the kernel with 32 x 32 blocks per wave over the upper triangle up to 8 pairs and the one with 64 x 64 blocks beyond.
Measured when the choice was made (one MI355X box; the 64 x 64 kernel forced for the first row):
  64 x 64 blocks:                         1 pair 34.6 us   2 pairs 34.3   4 pairs 35.0   8 pairs 36.9
  32 x 32 blocks, all 64 of them:         1 pair 14.4 us   2 pairs 14.5   4 pairs 24.1   8 pairs 34.5
  32 x 32 blocks, upper triangle (kept):  1 pair 14.7 us   2 pairs 14.6   4 pairs 15.4   8 pairs 25.5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd import ops
for P in (1, 2, 4, 8, 16):
    lengths = [382 - 7 * p for p in range(P)]; starts, r = [], 0
    for n in lengths: starts.append(r); r += (n + 31) // 32 * 32
    x = torch.randn(6, r, 256, device='cuda')
    f = lambda: ops._gram_per_pair(x, starts, lengths)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    ref = torch.stack([torch.stack([x[a, s:s + n].double().t() @ x[a, s:s + n].double() for s, n in zip(starts, lengths)]) for a in range(6)])
    print('pairs %2d  %.1f us  max err %.1e' % (P, e0.elapsed_time(e1) * 20, float((f().double() - ref).abs().max() / ref.abs().max())))
