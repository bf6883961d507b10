"""Equivariant cross attention of a batch at the bench shape (8 pairs, ~360 superpoints per cloud, A = 6, C = 256, H = 4): time per call of
ops.cross_attention_eq_stack (Gram products + statistics + the x6 apply kernel) and its error against the per-pair op.
python tools/micro/cross_eq_stack.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd import ops, tables
torch.manual_seed(0)
A, C, H = 6, 256, 4
lens_q = [382, 350, 361, 377, 340, 390, 365, 358]
lens_k = [350, 382, 377, 361, 390, 340, 358, 365]
def pack(lens):
    starts, r = [], 0
    for n in lens:
        starts.append(r); r += (n + 31) // 32 * 32
    return starts, r
sq, Rq = pack(lens_q); sk, Rk = pack(lens_k)
q = torch.randn(A, Rq, C, device='cuda') * 0.7
k = torch.randn(A, Rk, C, device='cuda') * 0.7
vt = torch.randn(A, C, Rk, device='cuda')
trace = torch.from_numpy(tables.trace_indices()[0]).long().cuda()
for mode in ('a_soft', 'r_soft'):
    out = torch.zeros_like(q)
    run = lambda: ops.cross_attention_eq_stack(q, k, vt, sq, lens_q, sk, lens_k, H, mode, trace, out)
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    err = 0.0
    for p in (0, 5):
        ref = ops.cross_attention_eq(q[:, sq[p]:sq[p] + lens_q[p]].contiguous(), k[:, sk[p]:sk[p] + lens_k[p]].contiguous(),
                                     torch.nn.functional.pad(vt[:, :, sk[p]:sk[p] + lens_k[p]], (0, ops.key_stride(lens_k[p]) - lens_k[p])).contiguous(), H, mode, trace)[0]
        err = max(err, float((out[:, sq[p]:sq[p] + lens_q[p]] - ref).abs().max() / ref.abs().max()))
    print('%s: %.1f us per call (all launches), max relative difference to the per-pair op %.1e' % (mode, a.elapsed_time(b) * 50, err))
