"""The transformer's dense-layer shapes (8 pairs per forward and one pair per forward): library GEMM (rocBLAS / hipBLASLt through torch) against
the tile kernel csrc/linear_f16.hip and the streaming kernel csrc/dense_norm.hip (plain mode).  python tools/micro/linear_stream_shapes.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, torch
import torch.nn.functional as F
from se3et_amd import ops
x = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < 1.0: y = x @ x
torch.cuda.synchronize()
def timeit(f, n=20):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (rows, K, N, bias, relu): 8 pairs: eq layers 6 x 5632 rows, cross directions 6 x 2816, invariant 5632 / 2816; one pair: / 8
shapes = []
for rows in (33792, 16896, 5632, 2816, 4224, 704, 352):
    shapes += [(rows, 256, 256, True, False), (rows, 256, 512, True, True), (rows, 512, 256, False, False)]
shapes += [(33792, 256, 1552, True, False), (4224, 256, 1552, True, False), (5632, 256, 1280, True, False), (704, 256, 1280, True, False),
           (33036, 1024, 256, True, False), (4130, 1024, 256, True, False), (5632, 1536, 512, True, True), (704, 1536, 512, True, True)]
tl = tf = ts = 0.0
for M, K, N, hb, relu in shapes:
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda') if hb else None
    lib_f = (lambda: torch._addmm_activation(b, a, w.t(), use_gelu=False)) if (relu and hb) else (lambda: F.linear(a, w, b))
    t_lib = timeit(lib_f)
    t_f16 = timeit(lambda: ops.linear_f16(a, w, b, relu))
    t_str = timeit(lambda: ops.linear_stream(a, w, b, relu))
    ref = F.linear(a.double(), w.double(), None if b is None else b.double())
    ref = ref.clamp_min(0) if relu else ref
    err = float((ops.linear_stream(a, w, b, relu).double() - ref).abs().max() / ref.abs().max())
    tl += t_lib; tf += t_f16; ts += t_str
    print('M %6d K %4d N %4d %s%s  library %6.1f us  tile kernel %6.1f us  streaming kernel %6.1f us  rel err %.1e' % (M, K, N, 'b' if hb else ' ', 'r' if relu else ' ', t_lib, t_f16, t_str, err))
print('sum library %.0f us  tile kernel %.0f us  streaming kernel %.0f us' % (tl, tf, ts))
# transposed value projection against baddbmm
for A, R in ((6, 5632), (6, 2816), (1, 5632), (6, 704), (1, 704)):
    x3 = torch.randn(A, R, 256, device='cuda'); w = torch.randn(256, 256, device='cuda') / 16; b = torch.randn(256, device='cuda')
    f_lib = lambda: torch.baddbmm(b[None, :, None].expand(A, 256, R), w[None].expand(A, 256, 256), x3.transpose(1, 2))
    t_lib, t_str = timeit(f_lib), timeit(lambda: ops.linear_stream_transposed(x3, w, b))
    err = float((ops.linear_stream_transposed(x3, w, b) - f_lib()).abs().max())
    print('V^T A %d R %5d  library baddbmm %6.1f us  streaming kernel (transposed store) %6.1f us  max diff %.1e' % (A, R, t_lib, t_str, err))
