"""GPU time of the KPConv products (P*6, 36*Cin) @ (36*Cin, Cout) in different library formulations (not a test)."""
import time, torch
import torch.nn.functional as F
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for P, Cin, Cout in [(80000, 16, 32), (51200, 32, 32), (51200, 32, 64), (21400, 64, 64), (21400, 64, 128), (5500, 128, 128), (5500, 128, 256), (10000, 16, 32), (6400, 32, 64)]:
    M, K, N = P * 6, 36 * Cin, Cout
    G = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda'); Wt = W.t().contiguous()
    res = {}
    for lib in ('cublaslt', 'cublas'):
        torch.backends.cuda.preferred_blas_library(lib)
        res[lib + ' mm(G,W)'] = t(lambda: torch.mm(G, W))
        res[lib + ' mm(G,Wt.t())'] = t(lambda: torch.mm(G, Wt.t()))
        res[lib + ' (Wt@G.t()).t()'] = t(lambda: torch.mm(Wt, G.t()))
    gb = (M * K + K * N + M * N) * 4 / 1e9
    print('P=%6d Cin=%3d Cout=%3d  M=%7d K=%5d N=%3d  %.2f GB %.1f GF : ' % (P, Cin, Cout, M, K, N, gb, 2 * M * K * N / 1e9) +
          ' | '.join('%s %.0f us' % kv for kv in res.items()))
