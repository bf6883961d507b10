"""Where the waves of the fused KPConv kernel spend a step (diagnostic build with s_memtime stamps): python tools/micro/kpconv_stamps.py <layer 0..9>
Producer slots: 0 step start, 1 next point's rows requested, 2 MFMAs + next operands requested, 3 rows split and stored, 4 barrier left.  Consumer: 0 chunk start, 3 MFMAs issued, 4 barrier left."""
import ctypes, os, sys; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
from se3et_amd import ops, tables, _lib
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair
L = ctypes.CDLL(os.path.join(R, 'tools/micro/' + os.environ.get('KPCONV_STAMPS_LIB', 'libkpconv_stamps.so')))
layer = int(sys.argv[1]); dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
dd = precompute_data_stack_mode(pts, torch.tensor([len(c) for c in clouds]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
calls_ = [(0, 0, 'neighbors', 32), (1, 0, 'subsampling', 32), (1, 1, 'neighbors', 64), (1, 1, 'neighbors', 64), (2, 1, 'subsampling', 64),
          (2, 2, 'neighbors', 128), (2, 2, 'neighbors', 128), (3, 2, 'subsampling', 128), (3, 3, 'neighbors', 256), (3, 3, 'neighbors', 256)]
qs, ss, tab, C = calls_[layer]
g = torch.Generator(device='cpu').manual_seed(0)
q, s = dd['points'][qs], dd['points'][ss]
idx = dd[tab][qs if tab == 'neighbors' else ss]
x = torch.randn(s.shape[0], 6, C, generator=g).to(dev)
w = (torch.randn(6, 6, C, C, generator=g) / (36 * C) ** 0.5).to(dev)
kp = torch.from_numpy(tables.kernel_points(b.init_radius * 2 ** ss)).to(dev)
sig = b.init_sigma * 2 ** ss
P, NN = idx.shape; vp = ctypes.c_void_p
Wp = ops._kpconv_weight_pieces(w, C, C, ops._stream())
out = torch.empty((P, 6, C), device=dev)
nbytes = _lib.lib().se3_kpconv_neighbor_table_bytes(P, NN)
ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
_lib.check(_lib.lib().se3_kpconv_neighbor_table(q.data_ptr(), s.data_ptr(), idx.data_ptr(), kp.data_ptr(), float(sig), P, s.shape[0], NN, ws.data_ptr(), nbytes, None), 'table')
stamps = torch.zeros((64, 16, 40, 8), dtype=torch.int64, device=dev)
L.se3_kpconv_so3_fused.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_size_t, ctypes.c_int, vp]
L.se3_debug_kpconv_set_stamps.argtypes = [vp]
assert L.se3_debug_kpconv_set_stamps(stamps.data_ptr()) == 0
torch.cuda.synchronize()
blocked = len(sys.argv) > 2 and sys.argv[2] == 'blocked' and C % 16 == 0
xin = x.view(s.shape[0], 3, 2, C // 16, 16).permute(0, 3, 1, 4, 2).contiguous() if blocked else x
for _ in range(3):
    rc = L.se3_kpconv_so3_fused(xin.data_ptr(), ws.data_ptr(), P, s.shape[0], NN, C, C, Wp.data_ptr(), out.data_ptr(), None, 0, 1 if blocked else 0, None)
    assert rc == 0
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.float64)
chunks = C // 8; steps = chunks + 2
nc = {32: 3, 64: 3, 128: 4, 256: 4}[C]
blk = st[:32]
prod = blk[:, nc:nc + 8, :min(steps, 40)]           # (blocks, waves, steps, slots)
ok = prod[..., 0] > 0
def seg(a, i, j, m): d = (a[..., j] - a[..., i])[m]; return d.mean(), np.percentile(d, 90)
act = ok & (prod[..., 2] > 0) & (prod[..., 1] > 0)
print('layer %d%s: P %d C %d, %d chunks; ticks of s_memtime (100 MHz realtime? -> treat as cycles of the shader clock)' % (layer, ' (blocked x)' if blocked else '', P, C, chunks))
for name, i, j in (('held rows + next requests', 0, 1), ('  (acc init, count)', 1, 5), ('  split operands + 18 MFMAs', 5, 6), ('  extra rounds (> 40 neighbours)', 6, 7), ('  next operand requests', 7, 2), ('split result + stores', 2, 3), ('barrier wait', 3, 4)):
    m, p90 = seg(prod, i, j, act & (prod[..., j] > 0)); print('  producer %-30s mean %8.0f  p90 %8.0f' % (name, m, p90))
m, p90 = seg(prod, 0, 4, act & (prod[..., 4] > 0)); print('  producer %-30s mean %8.0f  p90 %8.0f' % ('whole step', m, p90))
cons = blk[:, :nc, 2:min(steps, 40)]
okc = (cons[..., 0] > 0) & (cons[..., 3] > 0)
m, p90 = seg(cons, 0, 3, okc); print('  consumer %-30s mean %8.0f  p90 %8.0f' % ('MFMA chunk', m, p90))
m, p90 = seg(cons, 3, 4, okc & (cons[..., 4] > 0)); print('  consumer %-30s mean %8.0f  p90 %8.0f' % ('barrier wait', m, p90))
