"""Can the library's bf16 GEMMs carry the wide KPConv products at f32 accuracy?  out = G W with G = G1 + G2 + G3, W = W1 + W2 + W3 (bf16
pieces) as THREE library GEMMs with f32 output: G1 [W1 W2 W3], G2 [W1 W2], G3 W1, column blocks summed.  Prints rates and the error
against float64.  python tools/micro/bf16x6_library.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device('cuda')
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def split3(x):
    a = x.to(torch.bfloat16); r = x - a.float(); b = r.to(torch.bfloat16); c = (r - b.float()).to(torch.bfloat16)
    return a, b, c
for M, K, N in ((33036, 9216, 256), (128466, 4608, 128), (33036, 4608, 128)):
    g = torch.Generator().manual_seed(0)
    G = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)
    t32 = timeit(lambda: torch.mm(G, W))
    G1, G2, G3 = split3(G); W1, W2, W3 = split3(W)
    Wa = torch.cat((W1, W2, W3), 1).contiguous(); Wb = torch.cat((W1, W2), 1).contiguous()
    try:
        f = lambda: (torch.mm(G1, Wa, out_dtype=torch.float32), torch.mm(G2, Wb, out_dtype=torch.float32), torch.mm(G3, W1, out_dtype=torch.float32))
        f()
    except Exception as e:
        print('out_dtype not supported:', repr(e)[:200]); break
    def full():
        a, b, c = f()
        return (a[:, 2 * N:] + b[:, N:] + c) + (a[:, N:2 * N] + b[:, :N]) + a[:, :N]
    t6 = timeit(full)
    ta, tb, tc = timeit(lambda: torch.mm(G1, Wa, out_dtype=torch.float32)), timeit(lambda: torch.mm(G2, Wb, out_dtype=torch.float32)), timeit(lambda: torch.mm(G3, W1, out_dtype=torch.float32))
    rows = slice(0, 4096)
    ref = (G[rows].double() @ W.double())
    e32 = float((torch.mm(G, W)[rows].double() - ref).abs().max() / ref.abs().max())
    e6 = float((full()[rows].double() - ref).abs().max() / ref.abs().max())
    gf = 2.0 * M * K * N / 1e9
    print('M %6d K %5d N %3d: f32 library %.3f ms (%.0f TF/s)  bf16x6 library %.3f ms (3 GEMMs %.3f + %.3f + %.3f)  err f32 %.2e  bf16x6 %.2e'
          % (M, K, N, t32, gf / t32, t6, ta, tb, tc, e32, e6))
