"""GPU time of the geometric-embedding kernel at the bench shapes (not a test)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from se3et_amd import functional as SF, tables
g = torch.Generator().manual_seed(8)
C = 256
div = torch.exp(torch.arange(0, C, 2).float() * (-np.log(10000.0) / C)).cuda()
w = [(torch.randn(C, C, generator=g) / C ** 0.5).cuda() for _ in range(2)]
b = [(torch.randn(C, generator=g) * 0.1).cuda() for _ in range(2)]
w1 = torch.from_numpy(tables.wigner_tables()[1]).cuda()
for N in (382, 350, 263, 1404):
    pts = (torch.rand(N, 3, generator=g) * torch.tensor([1.5, 1.2, 1.0])).cuda()
    for dt in (torch.float32, torch.bfloat16):
        f = lambda: SF.geometric_embedding(pts, div, w[0], b[0], w[1], b[1], 0.2, 15.0, 3, wigner_d1=w1, dtype=dt)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print('N=%4d %s: %6.1f us per call (knn3 + embedding), %.2f TB/s of output' % (N, str(dt)[6:], us, N * N * C * (4 if dt == torch.float32 else 2) / us / 1e6))
