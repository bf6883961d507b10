"""The unary layers of the bench pyramid (8 pairs stacked): dense layer + GroupNorm as today (GEMM, then three GroupNorm launches) against the
fused kernel (csrc/dense_norm.hip: statistics in the epilogue) + the apply pass.  python tools/micro/dense_norm_shapes.py [target_chunks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from se3et_amd import functional as SF
from se3et_amd import ops
from se3et_amd._lib import lib

SHAPES = [(480000, 64, 32), (480000, 32, 128), (480000, 64, 128), (480000, 128, 32), (310452, 32, 128), (310452, 128, 64), (310452, 64, 256),
          (310452, 128, 256), (310452, 256, 64), (128466, 64, 256), (128466, 256, 128), (128466, 128, 512), (128466, 256, 512),
          (128466, 512, 128), (33036, 128, 512), (33036, 512, 256), (33036, 256, 1024), (33036, 512, 1024), (33036, 1024, 256)]


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
    if len(sys.argv) > 1:
        lib().se3_dense_norm_set_target_chunks(int(sys.argv[1]))
    torch.manual_seed(0)
    tot = [0.0, 0.0, 0.0]
    for rows, K, N in SHAPES:
        cuts = torch.linspace(0, rows // 6, 9).long() * 6
        seg = [int(c) for c in cuts]
        seg[-1] = rows
        x = torch.randn(rows, K, device='cuda')
        w = torch.randn(N, K, device='cuda') / K ** 0.5
        b = torch.randn(N, device='cuda')
        gw, gb = torch.rand(N, device='cuda') + 0.5, torch.randn(N, device='cuda')
        with torch.no_grad():
            def old():
                return ops.group_norm_rows(SF.linear(x, w), gw, gb, 32, 1e-5, 0.1, None, b, seg)

            def gemm():
                return ops.dense_norm(x, w, b, gw, gb, 32, 1e-5, seg)

            def new():
                p = gemm()
                p.slopes[-1] = 0.1
                return ops.group_norm_apply(p)
            err = float((old() - new()).abs().max())
            t_old, t_gemm, t_new = timed(old), timed(gemm), timed(new)
        mb = rows * (K + N) * 4 / 1e6
        tot[0] += t_old; tot[1] += t_gemm; tot[2] += t_new
        print('M %6d K %4d N %4d  today %.3f ms  fused gemm+stats %.3f ms (%.0f MB: %.2f TB/s)  + apply %.3f ms  max diff %.1e'
              % (rows, K, N, t_old, t_gemm, mb, mb / t_gemm / 1e3, t_new, err))
    print('sum today %.2f ms  fused gemm+stats %.2f ms  + apply %.2f ms' % tuple(tot))


if __name__ == '__main__':
    main()
