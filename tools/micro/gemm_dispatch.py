"""Host cost of a GEMM dispatch through torch on ROCm (not a test): python tools/micro/gemm_dispatch.py"""
import os, sys, time
import torch
import torch.nn.functional as F
x = torch.randn(6, 768, 256, device='cuda'); w = torch.randn(1552, 256, device='cuda'); b = torch.randn(1552, device='cuda')
x2 = torch.randn(768, 256, device='cuda'); w2 = torch.randn(256, 256, device='cuda'); b2 = torch.randn(256, device='cuda')
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    return host / n * 1e6, tot / n * 1e6
print('backend pref', os.environ.get('TORCH_BLAS_PREFER_HIPBLASLT'), torch.backends.cuda.preferred_blas_library())
for name, fn in [('linear 3D big', lambda: F.linear(x, w, b)), ('linear 2D small', lambda: F.linear(x2, w2, b2)),
                 ('linear 2D small nobias', lambda: F.linear(x2, w2)), ('mm small', lambda: torch.mm(x2, w2)),
                 ('addmm small', lambda: torch.addmm(b2, x2, w2.t())), ('addmm_act', lambda: torch._addmm_activation(b2, x2, w2.t())),
                 ('empty', lambda: torch.empty(768, 256, device='cuda')), ('add', lambda: x2 + x2)]:
    h, tot = t(fn)
    print('%-24s host %.1f us/call, wall %.1f us/call' % (name, h, tot))
for lib in ('cublas', 'cublaslt'):
    torch.backends.cuda.preferred_blas_library(lib)
    print('preferred_blas_library ->', torch.backends.cuda.preferred_blas_library())
    for name, fn in [('linear 3D big', lambda: F.linear(x, w, b)), ('linear 2D small', lambda: F.linear(x2, w2, b2)),
                     ('mm small', lambda: torch.mm(x2, w2)), ('addmm_act', lambda: torch._addmm_activation(b2, x2, w2.t()))]:
        h, tot = t(fn)
        print('%-24s host %.1f us/call, wall %.1f us/call' % (name, h, tot))
