"""Micro-benchmark of the stack-mode RPE self-attention kernels (not a test): python tools/bench_attention_stack.py"""
import sys; sys.path.insert(0, '.')
import random, time
import torch
from se3et_amd import ops
from se3et_amd._lib import lib

C, H = 256, 4


def setup(A, lengths, eq):
    g = torch.Generator(device='cuda').manual_seed(0)
    r = lambda *s: torch.randn(*s, device='cuda', generator=g)
    starts, R = [], 0
    for n in lengths:
        starts.append(R); R += (n + 31) // 32 * 32
    width = 2 * C + H * C + (4 * H if eq else 0)
    proj = r(A, R, width) * 0.3
    vt = r(A, C, R)
    embs = [r(n, n, C) for n in lengths]
    eqs = [r(A, n, n, 4) for n in lengths] if eq else None
    return proj, vt, embs, eqs, starts


def call(proj, vt, embs, eqs, starts, lengths, out):
    qe = proj[..., 2 * C + H * C:] if eqs is not None else None
    bias, offs = ops.rpe_bias_stack(proj[..., 2 * C:2 * C + H * C], qe, embs, eqs, starts, lengths, H)
    ops.attention_stack(proj[..., :C], proj[..., C:2 * C], vt, bias, offs, starts, lengths, starts, lengths, H, out, tag='rpe')
    return bias, offs


def reference(proj, vt, embs, eqs, starts, lengths):
    outs = []
    A = proj.shape[0]
    for c, n in enumerate(lengths):
        rows = proj[:, starts[c]:starts[c] + n].double()
        q = rows[..., :C].view(A, n, H, C // H); k = rows[..., C:2 * C].view(A, n, H, C // H)
        qp = rows[..., 2 * C:2 * C + H * C].view(A, n, H, C)
        s = torch.einsum('anhd,amhd->ahnm', q, k) + torch.einsum('anhc,nmc->ahnm', qp, embs[c].double())
        if eqs is not None:
            qe = rows[..., 2 * C + H * C:].view(A, n, H, 4)
            s = s + torch.einsum('anhe,anme->ahnm', qe, eqs[c].double())
        p = torch.softmax(s / (C // H) ** 0.5, -1)
        v = vt[:, :, starts[c]:starts[c] + n].double().view(A, H, C // H, n)
        outs.append(torch.einsum('ahnm,ahdm->anhd', p, v).reshape(A, n, C).float())
    return outs


def run(A, lengths, eq, bias_variant, split, attn_variant, iters=40, check=False):
    proj, vt, embs, eqs, starts = setup(A, lengths, eq)
    out = torch.zeros(A, proj.shape[1], C, device='cuda')
    lib().se3_debug_set_bias_variant(bias_variant, split)
    lib().se3_debug_set_attention_variant(attn_variant)
    for _ in range(3): call(proj, vt, embs, eqs, starts, lengths, out)
    if check:
        ref = reference(proj, vt, embs, eqs, starts, lengths)
        err = max(((out[:, s:s + n] - r).abs().max() / r.abs().max()).item() for s, n, r in zip(starts, lengths, ref))
        print('   check A=%d %s eq=%d bias_variant=%d split=%d attn_variant=%d: rel err %.2e' % (A, lengths, eq, bias_variant, split, attn_variant, err))
    ops.KERNEL_TIMINGS = {}
    for _ in range(iters): call(proj, vt, embs, eqs, starts, lengths, out)
    torch.cuda.synchronize()
    t = ops.KERNEL_TIMINGS; ops.KERNEL_TIMINGS = None
    f = lambda n: sum(a.elapsed_time(b) for a, b, _ in t[n]) / len(t[n]) * 1e3
    return f('rpe_bias_kernel'), f('attention_kernel@rpe'), t['rpe_bias_kernel'][0][2]


if __name__ == '__main__':
    x = torch.randn(4096, 4096, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 1.5: y = x @ x          # ramp the clocks
    torch.cuda.synchronize()
    shapes = [(6, (382, 350), True), (1, (382, 350), False), (6, (382,), True)]
    for A, lengths, eq in shapes:
        for bv, sp, av in ((0, 0, 0), (2, 2, 1), (0, 8, 2), (0, 1, 3), (2, 0, 4)):
            run(A, lengths, eq, bv, sp, av, iters=1, check=True)
    bias_cfgs = [(0, 0), (0, 2), (0, 6), (2, 2), (0, 3)]
    attn_cfgs = [0, 1, 2, 3, 4]
    res = {}
    for rep in range(3):
        order = [(s, b) for s in range(len(shapes)) for b in range(len(bias_cfgs))]
        random.shuffle(order)
        for si, bi in order:
            A, lengths, eq = shapes[si]
            av = attn_cfgs[bi % len(attn_cfgs)]
            tb, ta, nb = run(A, lengths, eq, bias_cfgs[bi][0], bias_cfgs[bi][1], av)
            res.setdefault(('bias', si, bias_cfgs[bi]), []).append(tb)
            res.setdefault(('attn', si, av), []).append(ta)
            res[('bytes', si)] = nb
    for si, (A, lengths, eq) in enumerate(shapes):
        nb = res[('bytes', si)]
        print('A=%d lengths=%s eq=%d  algorithmic %.1f MB' % (A, lengths, eq, nb / 1e6))
        for b in bias_cfgs:
            t = min(res[('bias', si, b)])
            print('   bias variant %d split %2d : %6.1f us' % (b[0], b[1], t))
        for av in attn_cfgs:
            t = min(res[('attn', si, av)])
            print('   attention variant %d    : %6.1f us' % (av, t))
        tb = min(min(res[('bias', si, b)]) for b in bias_cfgs); ta = min(min(res[('attn', si, av)]) for av in attn_cfgs)
        print('   best total %.1f us -> %.0f GB/s (%.1f%% of 8 TB/s)' % (tb + ta, nb / (tb + ta) / 1e3, nb / (tb + ta) / 1e3 / 80))
    lib().se3_debug_set_bias_variant(0, 0); lib().se3_debug_set_attention_variant(0)
