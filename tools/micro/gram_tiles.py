"""gram_stack at one pair and at eight pairs per forward, 64 x 64 blocks per wave against 32 x 32 (SE3_GRAM_TILE=64 / 32 forces one):
python tools/micro/gram_tiles.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for tile in ('64', '32'):
        subprocess.run([sys.executable, __file__, tile], env=dict(os.environ, SE3_GRAM_TILE=tile), check=True)
    sys.exit(0)
sys.path.insert(0, R)
import torch
from se3et_amd import ops
for P in (1, 2, 4, 8):
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
    print('tile %s  pairs %d  %.1f us  max err %.1e' % (sys.argv[1], P, e0.elapsed_time(e1) * 20, float((f().double() - ref).abs().max() / ref.abs().max())))
