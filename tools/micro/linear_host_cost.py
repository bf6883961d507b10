"""Host time per call of the ways to run y = x W^T + b on a small problem (the GPU finishes each before the host issues the next, so
wall time per call = host time per call): python tools/micro/linear_host_cost.py"""
import sys, time; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from se3et_amd import ops
x = torch.randn(2292, 256, device='cuda'); W = torch.randn(256, 256, device='cuda'); b = torch.randn(256, device='cuda'); Wt = W.t()
x3 = x.view(6, 382, 256)
cands = {
    'F.linear(x, W, b)': lambda: F.linear(x, W, b),
    'F.linear(x, W)': lambda: F.linear(x, W),
    'torch.addmm(b, x, W.t())': lambda: torch.addmm(b, x, W.t()),
    'torch.mm(x, W.t())': lambda: torch.mm(x, Wt),
    'torch.mm + add_': lambda: torch.mm(x, Wt).add_(b),
    'torch._addmm_activation(b, x, W.t())': lambda: torch._addmm_activation(b, x, Wt, use_gelu=False),
    'ops.linear_f16(x, W, b)': lambda: ops.linear_f16(x, W, b),
    'F.linear on (6, 382, 256)': lambda: F.linear(x3, W, b),
    'torch.empty': lambda: torch.empty((2292, 256), device='cuda'),
    'torch.cuda.current_stream()': lambda: torch.cuda.current_stream(),
    'ops._stream()': lambda: ops._stream(),
}
for name, f in cands.items():
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%-40s host %.1f us per call (drain afterwards %.1f ms)' % (name, (t1 - t0) / 2000 * 1e6, (t2 - t1) * 1e3))
