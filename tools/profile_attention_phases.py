"""Phase profile of attention_kernel (profiling hook, attention variant 9; not a test): python tools/profile_attention_phases.py
Per wave 32 clock64() stamps: 0 start, 1 after Q staging, per key tile t (2+4t ..): top, after S + logits, after softmax,
after P.V issue; 29 after the tile loop (= 30), 31 after merge + store."""
import sys; sys.path.insert(0, '.')
import importlib.util, ctypes, time
import torch
from se3et_amd import ops
from se3et_amd._lib import lib
spec = importlib.util.spec_from_file_location('bas', 'tools/bench_attention_stack.py'); bas = importlib.util.module_from_spec(spec); spec.loader.exec_module(bas)
C, H = 256, 4
x = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < 1.0: y = x @ x
torch.cuda.synchronize()
for A, lengths, eq in [(1, (382, 350), False), (6, (382, 350), True)]:
    proj, vt, embs, eqs, starts = bas.setup(A, lengths, eq)
    out = torch.zeros(A, proj.shape[1], C, device='cuda')
    qe = proj[..., 2 * C + H * C:] if eqs is not None else None
    bias, offs = ops.rpe_bias_stack(proj[..., 2 * C:2 * C + H * C], qe, embs, eqs, starts, lengths, H)
    G = len(lengths) * A * H
    QT = max((n + 31) // 32 for n in lengths)
    wgs = 8 * ((G + 7) // 8) * QT
    stamps = torch.zeros(wgs * 4 * 32, dtype=torch.int64, device='cuda')
    lib().se3_debug_set_attention_profile(ctypes.c_void_p(stamps.data_ptr()))
    lib().se3_debug_set_attention_variant(9)
    for _ in range(3):
        stamps.zero_()
        ops.attention_stack(proj[..., :C], proj[..., C:2 * C], vt, bias, offs, starts, lengths, starts, lengths, H, out)
    torch.cuda.synchronize()
    lib().se3_debug_set_attention_variant(0)
    s = stamps.view(wgs, 4, 32).cpu().double()
    live = s[:, 0, 0] > 0
    s = s[live]
    t_all = s[:, :, 31].max() - s[:, :, 0].min()
    print('A=%d: %d live workgroups; kernel span %.0f ticks' % (A, s.shape[0], t_all))
    d = lambda a, b: (s[:, :, b] - s[:, :, a]).mean().item()
    print('  Q staging %.0f | tile0: loads->S %.0f softmax %.0f PV-issue %.0f | tile1: top %.0f S %.0f softmax %.0f PV %.0f | tile2: top %.0f S %.0f softmax %.0f PV %.0f | loop end %.0f merge %.0f store %.0f | wave total %.0f' % (
        d(0, 1), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(6, 7), d(7, 8), d(8, 9), d(9, 10), d(10, 11), d(11, 12), d(12, 13), d(13, 29), d(29, 30), d(30, 31), d(0, 31)))
    start = s[:, :, 0] - s[:, :, 0].min()
    print('  wave start offsets: mean %.0f max %.0f ticks; wave end offsets: mean %.0f' % (start.mean(), start.max(), (s[:, :, 31] - s[:, :, 0].min()).mean()))
