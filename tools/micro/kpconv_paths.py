"""KPConv per stage at the bench shape (8 pairs per forward): round-1 path (slot sums G + library f32 GEMM) against the matrix-core
path (f16 hi / lo orbit sums + v_mfma_f32_32x32x16_f16 contraction), the latter also per stage.  python tools/micro/kpconv_paths.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd import ops, functional as SF, tables
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
kidx = torch.from_numpy(tables.kernel_slot_table()).to(dev); ridx = torch.from_numpy(tables.anchor_slot_table()).to(dev)
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# (query stage, support stage, table, Cin = Cout, sigma scale)
calls = [(0, 0, 'neighbors', 32), (1, 0, 'subsampling', 32), (1, 1, 'neighbors', 64), (1, 1, 'neighbors', 64), (2, 1, 'subsampling', 64),
         (2, 2, 'neighbors', 128), (2, 2, 'neighbors', 128), (3, 2, 'subsampling', 128), (3, 3, 'neighbors', 256), (3, 3, 'neighbors', 256)]
tot = {'old': 0.0, 'sums': 0.0, 'new': 0.0}
g = torch.Generator(device='cpu').manual_seed(0)
for qs, ss, tab, C in calls:
    q, s = dd['points'][qs], dd['points'][ss]
    idx = dd[tab][qs if tab == 'neighbors' else ss]
    x = torch.randn(s.shape[0], 6, C, generator=g).to(dev)
    w = (torch.randn(6, 6, C, C, generator=g) / (36 * C) ** 0.5).to(dev)
    kp = torch.from_numpy(tables.kernel_points(b.init_radius * 2 ** ss)).to(dev)
    sig = b.init_sigma * 2 ** ss
    res = {}
    for name, flag in (('old', False), ('sums', 'sums'), ('new', True)):
        ops.KPCONV_MATRIX_CORE = flag
        res[name] = timeit(lambda: SF.kpconv_inter_so3(x, q, s, idx, kp, w, kidx, ridx, sig))
        tot[name] += res[name]
    ops.KPCONV_MATRIX_CORE = 'auto'
    # the two stages of the matrix-core path by themselves
    from se3et_amd._lib import lib, check
    P, NN = idx.shape; st = ops._stream()
    Hs = torch.empty((lib().se3_kpconv_sums_bytes(P, C),), dtype=torch.uint8, device=dev)
    Wp = ops._kpconv_weight_pieces(w, C, C, st)
    out = torch.empty((P, 6, C), device=dev)
    nb = lib().se3_kpconv_neighbor_table_bytes(P, NN); tabl = torch.empty((nb,), dtype=torch.uint8, device=dev)
    t_t = timeit(lambda: check(lib().se3_kpconv_neighbor_table(q.data_ptr(), s.data_ptr(), idx.data_ptr(), kp.data_ptr(), float(sig), P, s.shape[0], NN, tabl.data_ptr(), nb, st), 't'))
    t_g = timeit(lambda: check(lib().se3_kpconv_so3_gather_sums(x.data_ptr(), tabl.data_ptr(), P, s.shape[0], NN, C, Hs.data_ptr(), st), 'g'))
    t_c = timeit(lambda: check(lib().se3_kpconv_so3_contract_f16(Hs.data_ptr(), Wp.data_ptr(), P, C, C, out.data_ptr(), st), 'c'))
    t_f = timeit(lambda: check(lib().se3_kpconv_so3_fused(x.data_ptr(), tabl.data_ptr(), P, s.shape[0], NN, C, C, Wp.data_ptr(), out.data_ptr(), None, 0, 0, st), 'f'))
    t_b = float('nan')
    if C % 16 == 0:      # the same kernel reading the blocked feature layout (whole cache lines per gather instruction)
        Ns = s.shape[0]
        xb = x.view(Ns, 3, 2, C // 16, 16).permute(0, 3, 1, 4, 2).contiguous()
        t_b = timeit(lambda: check(lib().se3_kpconv_so3_fused(xb.data_ptr(), tabl.data_ptr(), P, Ns, NN, C, C, Wp.data_ptr(), out.data_ptr(), None, 0, 1, st), 'f'))
        tot['blocked'] = tot.get('blocked', 0.0) + t_b
        check(lib().se3_kpconv_so3_fused(x.data_ptr(), tabl.data_ptr(), P, Ns, NN, C, C, Wp.data_ptr(), out.data_ptr(), None, 0, 0, st), 'f')
    t_s = float('nan')
    sb = lib().se3_kpconv_fused_split_workspace_bytes(P, C, C)
    if sb and C % 16 == 0:      # channels split over workgroups (round quantisation: few tiles, or a last round that is mostly empty)
        sws = torch.zeros((sb,), dtype=torch.uint8, device=dev)
        t_s = timeit(lambda: check(lib().se3_kpconv_so3_fused(xb.data_ptr(), tabl.data_ptr(), P, s.shape[0], NN, C, C, Wp.data_ptr(), out.data_ptr(), sws.data_ptr(), sb, 1, st), 'f'))
        tot['split'] = tot.get('split', 0.0) + t_s
        check(lib().se3_kpconv_so3_fused(x.data_ptr(), tabl.data_ptr(), P, s.shape[0], NN, C, C, Wp.data_ptr(), out.data_ptr(), None, 0, 0, st), 'f')
    elif C % 16 == 0:
        tot['split'] = tot.get('split', 0.0) + t_b
    err = float((out - torch.mm(ops.kpconv_slot_sums(x, q, s, idx, kp, kidx, ridx, sig), w.reshape(36 * C, C)).view(P, 6, C)).abs().max() / out.abs().max())
    gf = 2.0 * 6 * q.shape[0] * 36 * C * C / 1e9
    ops.KPCONV_MATRIX_CORE = True; outf = SF.kpconv_inter_so3(x, q, s, idx, kp, w, kidx, ridx, sig); ops.KPCONV_MATRIX_CORE = 'auto'
    errf = float((outf - torch.mm(ops.kpconv_slot_sums(x, q, s, idx, kp, kidx, ridx, sig), w.reshape(36 * C, C)).view(P, 6, C)).abs().max() / outf.abs().max())
    print('P %6d NN %2d C %3d  gemm %.3f ms  sums %.3f ms (table %.3f + gather %.3f + contract %.3f)  fused %.3f ms (kernel %.3f, blocked x %.3f, + channel split %.3f) (%.0f GF: %.0f TF/s f32-equivalent, %.2f PF/s f16)  err %.1e / %.1e'
          % (q.shape[0], idx.shape[1], C, res['old'], res['sums'], t_t, t_g, t_c, res['new'], t_f, t_b, t_s, gf, gf / res['new'], 3 * gf / res['new'] / 1e3, err, errf))
print('total gemm %.2f ms  sums %.2f ms  fused %.2f ms (kernels with blocked x: %.2f ms; with the channel split where it applies: %.2f ms) per 8 pairs' % (tot['old'], tot['sums'], tot['new'], tot.get('blocked', float('nan')), tot.get('split', float('nan'))))
