"""Runs the RPE self-attention kernels a few times (for rocprofv3 --pmc passes): python3 tests/pmc_attention.py"""
import sys; sys.path.insert(0, '.')
import torch
from se3et_amd import functional as SF
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
for (A, N, eq) in ((6, 382, True), (1, 382, False)):
    C, H = 256, 4
    q, k, v = r(A, N, C), r(A, N, C), r(A, N, C)
    emb = r(N, N, C); eqe = r(A, N, N, 4) if eq else None
    wp, weq = r(C, C) / 16, (r(C, 4) if eq else None)
    if A == 1: q, k, v = q[0], k[0], v[0]
    vt = SF.project_values_transposed(v, torch.eye(C, device='cuda'), torch.zeros(C, device='cuda'))
    for _ in range(5): SF.rpe_attention(q, k, vt, emb, wp, eqe, weq, H, False)
torch.cuda.synchronize()
