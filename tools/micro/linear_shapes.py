"""The dense-layer shapes of one 8-pair bench step: library f32 GEMM against csrc/linear_f16.hip.  python tools/micro/linear_shapes.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, torch
from se3et_amd import ops
x = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < 1.0: y = x @ x
torch.cuda.synchronize()
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(34560, 256, 1552), (128466, 512, 128), (128466, 128, 512), (33036, 256, 1024), (128466, 256, 512), (33036, 512, 1024), (21411, 1536, 512),
          (310452, 64, 256), (310452, 256, 64), (33036, 1024, 256), (310452, 128, 256), (34560, 256, 1536), (51742, 768, 256), (34560, 512, 256),
          (15360, 256, 256), (19200, 256, 256), (480000, 64, 128), (128466, 256, 128), (480000, 32, 128), (310452, 32, 128), (310452, 128, 64)]
tl = tn = 0.0
for M, K, N in shapes:
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5
    t_lib = timeit(lambda: ops.mm(a, w.t()))
    t_new = timeit(lambda: ops.linear_f16(a, w))
    err = float((ops.linear_f16(a, w) - ops.mm(a, w.t())).abs().max() / ops.mm(a, w.t()).abs().max())
    gf = 2.0 * M * K * N / 1e9
    tl += t_lib; tn += t_new
    print('M %6d K %4d N %4d  library %.3f ms (%.0f TF/s)  f16 split %.3f ms (%.0f TF/s)  bytes %.0f MB  rel diff %.1e' % (M, K, N, t_lib, gf / t_lib, t_new, gf / t_new, 4.0 * (M * K + M * N) / 1e6, err))
print('sum library %.2f ms  f16 split %.2f ms' % (tl, tn))
