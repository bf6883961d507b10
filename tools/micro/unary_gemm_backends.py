"""GPU time of the backbone / transformer dense layers (x (M, K) @ W^T (K, N)) by BLAS backend and weight layout (not a test)."""
import time, torch
import torch.nn.functional as F
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
shapes = [(310452, 64, 256), (310452, 256, 64), (310452, 128, 256), (310452, 128, 64), (128466, 128, 512), (128466, 512, 128),
          (128466, 256, 512), (128466, 256, 128), (33036, 256, 1024), (33036, 1024, 256), (33036, 512, 1024), (480000, 128, 32),
          (480000, 64, 128), (480000, 32, 128), (34560, 256, 1552), (21411, 1536, 512), (51742, 768, 256)]
for M, K, N in shapes:
    x = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda'); Wt = W.t().contiguous()
    res = {}
    for lib in ('cublaslt', 'cublas'):
        torch.backends.cuda.preferred_blas_library(lib)
        res[lib[2:] + ' x@W.t()'] = t(lambda: torch.mm(x, W.t()))
        res[lib[2:] + ' x@Wt'] = t(lambda: torch.mm(x, Wt))
    ideal = max((M * K + K * N + M * N) * 4 / 6.3e12, 2.0 * M * K * N / 140e12) * 1e6
    best = min(res, key=res.get)
    print('M=%6d K=%4d N=%4d ideal %5.0f us : ' % (M, K, N, ideal) + ' | '.join('%s %4.0f' % kv for kv in res.items()) + '  -> best %s (%.1fx ideal)' % (best, res[best] / ideal))
