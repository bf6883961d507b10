"""Batched Gram products X^T X of (48, W, 256) windows: time by window length and BLAS backend (not a test)."""
import time, torch
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for W in (304, 320, 352, 384, 389, 416, 448, 512):
    x = torch.randn(48, W, 256, device='cuda')
    res = {}
    for lib in ('cublaslt', 'cublas'):
        torch.backends.cuda.preferred_blas_library(lib)
        res[lib[2:] + ' bmm(x^T, x)'] = t(lambda: torch.bmm(x.transpose(1, 2), x))
        xt = x.transpose(1, 2).contiguous()
        res[lib[2:] + ' bmm(xt, x)'] = t(lambda: torch.bmm(xt, x))
    print('W=%3d: ' % W + ' | '.join('%s %5.1f' % kv for kv in res.items()))
