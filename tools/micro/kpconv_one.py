"""One KPConv shape on the matrix-core path, a few calls (workload for rocprofv3 --pmc passes): python tools/micro/kpconv_one.py P C [calls]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from se3et_amd import ops, functional as SF, tables
P, C = int(sys.argv[1]), int(sys.argv[2]); calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda'); g = torch.Generator().manual_seed(0)
pts = (torch.rand(P, 3, generator=g) * 2.0).to(dev)
d = torch.cdist(pts[:2048], pts)              # neighbours of the first 2048 points, tiled over all queries (timing only)
idx = d.topk(36, dim=1, largest=False)[1].repeat((P + 2047) // 2048, 1)[:P].contiguous()
x = torch.randn(P, 6, C, generator=g).to(dev); w = (torch.randn(6, 6, C, C, generator=g) / (36 * C) ** 0.5).to(dev)
kidx = torch.from_numpy(tables.kernel_slot_table()).to(dev); ridx = torch.from_numpy(tables.anchor_slot_table()).to(dev)
kp = torch.from_numpy(tables.kernel_points(0.5)).to(dev)
for _ in range(calls):
    y = SF.kpconv_inter_so3(x, pts, pts, idx, kp, w, kidx, ridx, 0.4)
torch.cuda.synchronize()
print('ok', tuple(y.shape))
