"""The tails of the ten bottleneck blocks of the bench pyramid (8 pairs stacked): round 3 (unary2 / skip_conv store their raw output + statistics,
one apply pass adds them) against round 4 (statistics-only GEMMs, then ONE kernel that runs the products again and writes the block's output:
csrc/dense_norm.hip MODE 1 / 2 / 3).  python tools/micro/block_tail.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from se3et_amd import ops

# (rows = points x 6 anchors, mid, out, skip_conv input width or 0 for an identity shortcut)
BLOCKS = [('1_2', 480000, 32, 128, 64), ('2_1', 310452, 32, 128, 0), ('2_2', 310452, 64, 256, 128), ('2_3', 310452, 64, 256, 0),
          ('3_1', 128466, 64, 256, 0), ('3_2', 128466, 128, 512, 256), ('3_3', 128466, 128, 512, 0), ('4_1', 33036, 128, 512, 0),
          ('4_2', 33036, 256, 1024, 512), ('4_3', 33036, 256, 1024, 0)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    torch.manual_seed(0)
    tot = [0.0, 0.0, 0.0, 0.0]
    for name, rows, mid, out, kin in BLOCKS:
        cuts = torch.linspace(0, rows // 6, 9).long() * 6
        seg = [int(c) for c in cuts]
        seg[-1] = rows
        dev = 'cuda'
        y = torch.randn(rows, mid, device=dev)
        st = [(torch.randn(8, 2, mid, device=dev) * 0.3 + torch.tensor([1.0, 0.0], device=dev)[None, :, None]) for _ in range(2)]
        pend = ops.Pending(y, st, [0.1, 0.1], seg)
        w2, b2 = torch.randn(out, mid, device=dev) / mid ** 0.5, torch.randn(out, device=dev)
        gw, gb = torch.rand(out, device=dev) + 0.5, torch.randn(out, device=dev)
        if kin:
            xs = torch.randn(rows, kin, device=dev)
            ws, bs = torch.randn(out, kin, device=dev) / kin ** 0.5, torch.randn(out, device=dev)
            gws, gbs = torch.rand(out, device=dev) + 0.5, torch.randn(out, device=dev)
        else:
            res = torch.randn(rows, out, device=dev)
        with torch.no_grad():
            def old():
                p = ops.dense_norm(pend, w2, b2, gw, gb, 32, 1e-5, seg)
                if kin:
                    return ops.group_norm_apply(p, ops.dense_norm(ops.Pending(xs, [], [], seg), ws, bs, gws, gbs, 32, 1e-5, seg), 0.1)
                return ops.group_norm_apply(p, res, 0.1)

            def stats():
                a = ops.dense_stats(pend, w2, b2, gw, gb, 32, 1e-5, seg)
                return (a, ops.dense_stats(xs, ws, bs, gws, gbs, 32, 1e-5, seg)) if kin else (a, None)

            affs = stats()

            def final():
                if kin:
                    return ops.dense_residual(pend, w2, affs[0], shortcut=(xs, ws, affs[1]), final_slope=0.1, segments=seg)
                return ops.dense_residual(pend, w2, affs[0], residual=res, final_slope=0.1, segments=seg)
            err = float((old() - final()).abs().max())
            t_old, t_stats, t_final = timed(old), timed(stats), timed(final)
        mb_new = rows * 4 * (2 * mid + (2 * kin if kin else out) + out) / 1e6           # y twice, the shortcut's input twice / the residual, the output
        tot[0] += t_old; tot[1] += t_stats; tot[2] += t_final; tot[3] += t_stats + t_final
        print('block %s  rows %6d  %4d -> %4d  %s  round 3 %.3f ms | statistics %.3f + final %.3f = %.3f ms (%.0f MB: %.2f TB/s)  max diff %.1e'
              % (name, rows, mid, out, 'skip_conv %4d' % kin if kin else 'identity     ', t_old, t_stats, t_final, t_stats + t_final, mb_new,
                 mb_new / (t_stats + t_final) / 1e3, err))
    print('sum round 3 %.2f ms | statistics %.2f + final %.2f = %.2f ms' % tuple(tot))


if __name__ == '__main__':
    main()
