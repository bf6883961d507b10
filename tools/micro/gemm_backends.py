"""GPU time + host dispatch cost of the path's GEMM shapes through hipBLASLt (F.linear with bias) and rocBLAS (mm), not a test."""
import time, torch
import torch.nn.functional as F
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    return host / n * 1e6, tot / n * 1e6
shapes = [(768, 256, 256), (768, 256, 512), (768, 512, 256), (4608, 256, 256), (4608, 256, 1552), (768, 256, 1040), (4608, 256, 512),
          (4608, 512, 256), (768, 1024, 256), (60000, 32, 32), (15000, 128, 64), (3600, 512, 256), (60000, 576, 32), (15000, 2304, 128)]
for M, K, N in shapes:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda'); wt = w.t().contiguous()
    torch.backends.cuda.preferred_blas_library('cublaslt')
    h1, g1 = t(lambda: F.linear(x, w, b))
    torch.backends.cuda.preferred_blas_library('cublas')
    h2, g2 = t(lambda: torch.mm(x, w.t()))
    h3, g3 = t(lambda: torch.mm(x, wt))
    print('M=%6d K=%5d N=%5d  Lt linear+bias: host %5.1f wall %6.1f us | rocBLAS mm(x, w.t()): host %5.1f wall %6.1f | mm(x, wt): host %5.1f wall %6.1f' % (M, K, N, h1, g1, h2, g2, h3, g3))
