"""GPU time of the KPConv gather kernel alone on the 11 convolutions of one 8-pair forward (not a test)."""
import sys; sys.path.insert(0, '.')
import ctypes, numpy as np, torch
from se3et_amd import ops as _ops
from se3et_amd._lib import lib, check
exec(open('tests/micro/kpconv_chunk_l3.py').read().split("def run(a, chunk_rows, Gbuf):")[0])
def gather(a, G):
    x, q_pts, s_pts, idx, kernel_points, weights, kidx, ridx, sigma = a
    x, q_pts, s_pts, idx = x.contiguous(), q_pts.contiguous(), s_pts.contiguous(), idx.contiguous()
    P, NN = idx.shape; Ns, A, Cin = x.shape
    kp, kt, rt = _ops._host_table(kernel_points, torch.float32), _ops._host_table(kidx, torch.int64), _ops._host_table(ridx, torch.int64)
    check(lib().se3_kpconv_so3_gather(q_pts.data_ptr(), s_pts.data_ptr(), idx.data_ptr(), x.data_ptr(), kp.data_ptr(), kt.data_ptr(), rt.data_ptr(), float(sigma), P, Ns, NN, Cin, G.data_ptr(), _ops._stream()), 'g')
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.set_grad_enabled(False)
for a in calls:
    idx, x = a[3], a[0]
    P, NN = idx.shape; Cin = x.shape[2]
    G = torch.empty(P * 6 * 36 * Cin, device='cuda')
    line = 'P=%6d NN=%2d Cin=%3d G=%5.0f MB:' % (P, NN, Cin, G.numel() * 4 / 1e6)
    us = t(lambda: gather(a, G))
    line += ' %5.0f us  (%.2f TB/s of G written)' % (us, G.numel() * 4 / us / 1e6)
    print(line, flush=True)
