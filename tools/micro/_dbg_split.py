import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from se3et_amd import ops, functional as SF
from test_gpu_ops import _conv_state
for (P, Ns, NN, Cin, Cout) in [(300, 800, 36, 64, 64)]:
    g = torch.Generator().manual_seed(1)
    s_pts = torch.rand(Ns, 3, generator=g) * 0.1
    q_pts = s_pts[:P].contiguous()
    d = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    idx = d.topk(NN, dim=1, largest=False)[1]
    x = torch.randn(Ns, 6, Cin, generator=g)
    st = _conv_state(Cin, Cout, 0.0625)
    def run(xx):
        return SF.kpconv_inter_so3(xx.cuda(), q_pts.cuda(), s_pts.cuda(), idx.cuda(), st['kernel_points'].cuda(), st['weights'].cuda(), st['kidx_rot'][:, 0, :].cuda(), st['ridx_rot'][0].cuda(), 0.05).cpu()
    a = run(x)
    ops.KPCONV_SPLIT = False
    b = run(x)
    x0 = x.clone(); x0[:, :, Cin // 2:] = 0
    x1 = x.clone(); x1[:, :, :Cin // 2] = 0
    b0, b1 = run(x0), run(x1)
    ops.KPCONV_SPLIT = True
    for p in (0, 1, 2, 8, 9, 17):
        print('point', p, 'a-b %.3g' % float((a[p] - b[p]).abs().max()), 'a-b0 %.3g' % float((a[p] - b0[p]).abs().max()), 'a-b1 %.3g' % float((a[p] - b1[p]).abs().max()),
              'a-2b0 %.3g' % float((a[p] - 2 * b0[p]).abs().max()), 'a-2b1 %.3g' % float((a[p] - 2 * b1[p]).abs().max()), '|b| %.3g' % float(b[p].abs().max()))
    # which point's value does a[1] hold?
    for p in (1, 2, 9):
        dist = [(float((a[p] - b[q]).abs().max()), q) for q in range(0, 32)]
        print('a[%d] closest to b[q]:' % p, sorted(dist)[:3])
    a = run(x)
    ws = list(ops._kpconv_split_ws.values())[0]
    tiles = (P + 15) // 16
    NCT = Cout // 32
    off = (tiles * NCT * 4 + 255) // 256 * 256
    part = ws[off:off + 2 * tiles * 16 * 6 * Cout * 4].view(torch.float32).view(2, tiles * 16, 6, Cout).cpu()
    inv = None
    for z, bz in ((0, b0), (1, b1)):
        pz = part[z][:P]
        scale = float((bz[0] / pz[0]).flatten()[0])
        print('z', z, 'scale', scale, 'point0 err %.3g' % float((pz[0] * scale - bz[0]).abs().max()), 'point1 err %.3g' % float((pz[1] * scale - bz[1]).abs().max()),
              'point1 vs b[0] %.3g' % float((pz[1] * scale - bz[0]).abs().max()), 'counters', ws[:off].view(torch.int32).abs().sum().item())
