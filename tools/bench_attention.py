"""Micro-benchmark of the RPE self-attention kernels (not a test): python tools/bench_attention.py"""
import sys; sys.path.insert(0, '.')
import math, torch
from se3et_amd import ops
from se3et_amd._lib import lib

def run(A, N, C, H, eq, variant, split, iters=30):
    g = torch.Generator(device='cuda').manual_seed(0)
    r = lambda *s: torch.randn(*s, device='cuda', generator=g)
    q, k, v = r(A, N, C), r(A, N, C), r(A, N, C)
    emb = r(N, N, C); eqe = r(A, N, N, 4) if eq else None
    wp, weq = r(C, C) / 16, (r(C, 4) if eq else None)
    lib().se3_debug_set_bias_variant(variant, split)
    if A == 1: q, k, v = q[0], k[0], v[0]
    from se3et_amd import functional as SF
    v = SF.project_values_transposed(v, torch.eye(C, device='cuda'), torch.zeros(C, device='cuda'))
    for _ in range(3): SF.rpe_attention(q, k, v, emb, wp, eqe, weq, H, False)
    ops.KERNEL_TIMINGS = {}
    for _ in range(iters): SF.rpe_attention(q, k, v, emb, wp, eqe, weq, H, False)
    torch.cuda.synchronize()
    t = ops.KERNEL_TIMINGS; ops.KERNEL_TIMINGS = None
    f = lambda n: sum(a.elapsed_time(b) for a, b, _ in t[n]) / len(t[n]) * 1e3
    nb = t['rpe_bias_kernel'][0][2]
    return f('rpe_bias_kernel'), f('rpe_bias_kernel') + f('attention_kernel@rpe'), nb, f('attention_kernel@rpe')

import random, time
x = torch.randn(4096, 4096, device='cuda')
t0 = time.time()
while time.time() - t0 < 1.5: y = x @ x          # ramp the clocks
torch.cuda.synchronize()
configs = [(A, N, eq, v, sp) for (A, N, eq) in ((6, 382, True), (6, 304, True), (1, 382, False)) for v in (0, 1, 2, 3, 4) for sp in (2,)]
res = {}
for rep in range(3):
    random.shuffle(configs)
    for c in configs:
        A, N, eq, v, sp = c
        lib().se3_debug_set_attention_variant(v)
        res.setdefault(c, []).append(run(A, N, 256, 4, eq, 0, sp, iters=60))
for c in sorted(res):
    tb = min(r[0] for r in res[c]); ta = min(r[1] for r in res[c]); nb = res[c][0][2]; tk = min(r[3] for r in res[c])
    print('A=%d N=%d eq=%d attn_variant=%d  bias %.1f us  attn %.1f us  total %.1f us  -> %.0f GB/s (%.1f%% of 8TB/s)' % (c[0], c[1], c[2], c[3], tb, tk, ta, nb / ta / 1e3, nb / ta / 1e3 / 80))
lib().se3_debug_set_bias_variant(0, 0); lib().se3_debug_set_attention_variant(0)
