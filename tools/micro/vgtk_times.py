"""Kernel time of the EPN-toolkit ops (se3et_amd.vgtk, csrc/vgtk_ops.hip) at the sizes the toolkit's own models use them (2 clouds of 1 024
points, 60 anchors, 12 kernel points in two rings, 32 neighbours, 64 channels): python tools/micro/vgtk_times.py
These ops are not on the SE3ET path; the numbers are evidence that the kernels are usable, not tuned."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd import vgtk
dev = 'cuda'; g = torch.Generator().manual_seed(0)
b, n, m, c, na, ks, nn_ = 2, 1024, 256, 64, 60, 12, 32
pts = torch.rand(b, 3, n, generator=g).to(dev)
feats = torch.randn(b, c, n, generator=g).to(dev)
idx = torch.randint(0, n, (b, m), generator=g).int().to(dev)
anchors = torch.nn.functional.normalize(torch.randn(na, 3, generator=g), dim=1).to(dev)
kernel_points = torch.rand(ks, 2, generator=g).to(dev)
grouped = (torch.rand(b, 3, m, nn_, generator=g) * 0.2 - 0.1).to(dev)
kernels3 = (torch.randn(ks, na, 3, generator=g) * 0.1).to(dev)
inter_idx = torch.randint(0, n, (b, m, na, ks, 4), generator=g).int().to(dev)
inter_w = torch.rand(b, m, na, ks, 4, generator=g).to(dev)
feats4 = torch.randn(b, c, n, na, generator=g).to(dev)
intra_idx = torch.randint(0, na, (b, m, na, 4), generator=g).int().to(dev) if False else None
def t(name, f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    print('%-46s %8.1f us' % (name, e0.elapsed_time(e1) / reps * 1e3))
t('gather_points (2, 64, 1024) -> 256', lambda: vgtk.gather_points(feats, idx))
fg = feats.clone().requires_grad_(True)
def gather_bwd():
    out = vgtk.Gathering.apply(fg, idx); out.backward(torch.ones_like(out)); fg.grad = None
t('gather_points forward + backward', gather_bwd)
t('ball_query_index 256 queries, 1024 points, 32', lambda: vgtk.ball_query_index(pts[:, :, :m].contiguous(), pts, 0.2, nn_))
t('furthest_sample_index 1024 -> 256', lambda: vgtk.furthest_sample_index(pts, m))
t('anchor_query (2, 3, 256, 32) x 60 anchors x 12', lambda: vgtk.anchor_query(None, None, grouped, anchors, kernel_points))
t('initial_anchor_query 1024 points, 256 centers', lambda: vgtk.initial_anchor_query(pts[0].t().contiguous(), pts[:, :, :m].contiguous(), kernels3, 0.2, 0.05))
t('inter_zpconv_grouping (2, 64, 1024, 60) -> 256', lambda: vgtk.inter_zpconv_grouping(inter_idx, inter_w, feats4))
f4 = feats4.clone().requires_grad_(True)
def inter_bwd():
    out = vgtk.inter_zpconv_grouping(inter_idx, inter_w, f4); out.backward(torch.ones_like(out)); f4.grad = None
t('inter_zpconv_grouping forward + backward', inter_bwd, reps=5)
