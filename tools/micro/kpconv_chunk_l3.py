"""Experiment (not a test): does the (P*6, 36*Cin) KPConv operand stay in the 256 MiB Infinity Cache when the gather + GEMM run in
row chunks that reuse ONE small G buffer?  Captures the KPConv calls of one 8-pair forward and replays each whole vs chunked."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from se3et_amd import ops as _ops
from se3et_amd._lib import lib, check
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
from se3et_amd.batched import forward_pairs

dev = torch.device('cuda')
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
clouds = []
for j in range(8):
    ref, src, _ = make_pair('c2_5k', index=j)
    clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
lens = torch.tensor([len(c) for c in clouds], dtype=torch.int64)
b = cfg.backbone
data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
data['features'] = torch.ones((pts.shape[0], 1), device=dev)

calls = []
orig = _ops.kpconv_inter_so3
def rec(*a):
    calls.append(a)
    return orig(*a)
_ops.kpconv_inter_so3 = rec
with torch.no_grad():
    forward_pairs(model, data)
_ops.kpconv_inter_so3 = orig


def run(a, chunk_rows, Gbuf):
    x, q_pts, s_pts, idx, kernel_points, weights, kidx, ridx, sigma = a
    x, q_pts, s_pts, idx = x.contiguous(), q_pts.contiguous(), s_pts.contiguous(), idx.contiguous()
    assert x.dtype == torch.float32 and idx.dtype == torch.int64 and q_pts.shape[0] == idx.shape[0] and s_pts.shape[0] == x.shape[0]
    P, NN = idx.shape
    Ns, A, Cin = x.shape
    Cout = weights.shape[-1]
    kp, kt, rt = _ops._host_table(kernel_points, torch.float32), _ops._host_table(kidx, torch.int64), _ops._host_table(ridx, torch.int64)
    W = weights.reshape(36 * Cin, Cout)
    out = torch.empty((P * 6, Cout), device=dev)
    st = _ops._stream()
    for p0 in range(0, P, chunk_rows):
        n = min(chunk_rows, P - p0)
        G = Gbuf[:n * 6 * 36 * Cin].view(n * 6, 36 * Cin)
        check(lib().se3_kpconv_so3_gather(q_pts[p0:].data_ptr(), s_pts.data_ptr(), idx[p0:].data_ptr(), x.data_ptr(), kp.data_ptr(),
                                          kt.data_ptr(), rt.data_ptr(), float(sigma), n, Ns, NN, Cin, G.data_ptr(), st), 'gather')
        torch.mm(G, W, out=out[p0 * 6:(p0 + n) * 6])
    return out


def gpu_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.backends.cuda.preferred_blas_library('cublaslt')
torch.set_grad_enabled(False)
for a in calls:
    x, idx, weights = a[0], a[3], a[5]
    P, NN = idx.shape
    Cin, Cout = x.shape[2], weights.shape[-1]
    full = P * 6 * 36 * Cin
    Gbuf = torch.empty(full, device=dev)
    ref = run(a, P, Gbuf)
    line = 'P=%6d NN=%2d Cin=%3d Cout=%3d G=%5.0f MB: whole %6.0f us' % (P, NN, Cin, Cout, full * 4 / 1e6, gpu_time(lambda: run(a, P, Gbuf)))
    for mb in (128, 64, 32, 16):
        rows = max(256, int(mb * 1e6 / (6 * 36 * Cin * 4)) // 256 * 256)
        if rows >= P:
            continue
        out = run(a, rows, Gbuf)
        assert torch.equal(out, ref) or (out - ref).abs().max() < 1e-3 * ref.abs().max(), 'mismatch'
        line += ' | %3d MB %6.0f us' % (mb, gpu_time(lambda: run(a, rows, Gbuf)))
    print(line, flush=True)
    del Gbuf
