"""Phase profile of attention_x6_kernel (profiling hook, attention variant 12; not a test): python tools/profile_attention_x6_phases.py
Per wave clock64() stamps: 0 start, 1 after the prologue (first tile published), per key tile t < 5 (2 + 5 t ..): after the requests,
after q.k + masking, after softmax + P split, after P.v issue, after publish + barrier; 30 after the loop."""
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
for A, lengths, eq in [(1, (382, 350), False), (6, (382, 350) * 8, True)]:
    proj, vt, embs, eqs, starts = bas.setup(A, lengths, eq)
    out = torch.zeros(A, proj.shape[1], C, device='cuda')
    qe = proj[..., 2 * C + H * C:] if eqs is not None else None
    bias, offs = ops.rpe_bias_stack(proj[..., 2 * C:2 * C + H * C], qe, embs, eqs, starts, lengths, H)
    wgs = ((max(lengths) + 127) // 128) * H * len(lengths) * A
    stamps = torch.zeros(wgs * 4 * 32, dtype=torch.int64, device='cuda')
    lib().se3_debug_set_attention_profile(ctypes.c_void_p(stamps.data_ptr()))
    lib().se3_debug_set_attention_variant(12)
    for _ in range(3):
        stamps.zero_()
        ops.attention_stack(proj[..., :C], proj[..., C:2 * C], vt, bias, offs, starts, lengths, starts, lengths, H, out)
    torch.cuda.synchronize()
    lib().se3_debug_set_attention_variant(0)
    s = stamps.view(wgs, 4, 32).cpu().double()
    s = s[s[:, 0, 1] > 0][:, 0:3]              # waves 0..2 (the fourth wave of the last query block may be idle)
    d = lambda a, b: (s[:, :, b] - s[:, :, a]).mean().item()
    print('A=%d, %d clouds: %d live workgroups; span %.0f ticks (clock64: 100 MHz -> x24 for 2.4 GHz cycles)' % (A, len(lengths), s.shape[0], (s[:, :, 30].max() - s[:, :, 0].min())))
    print('  prologue %.0f' % d(0, 1))
    for t in range(5):
        b = 2 + 5 * t
        print('  tile %d: requests %.0f | q.k + mask %.0f | softmax + split %.0f | P.v %.0f | publish + barrier %.0f' % (
            t, d(b - 1 if t else 1, b), d(b, b + 1), d(b + 1, b + 2), d(b + 2, b + 3), d(b + 3, b + 4)))
    print('  loop total %.0f ticks for %d tiles' % (d(1, 30), (max(lengths) + 31) // 32))
