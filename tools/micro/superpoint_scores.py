"""se3_superpoint_scores_stack at the bench shape (8 pairs of 382 x 304 superpoints, 256 channels): time per call and a checksum of the
scores (A/B of two libraries: SE3_LIB=...).    python tools/micro/superpoint_scores.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hashlib, torch
from se3et_amd import ops
g = torch.Generator().manual_seed(0)
B, N, M, C = 8, 382, 304, 256
feats = torch.nn.functional.normalize(torch.randn(B * (N + M), C, generator=g), dim=1).cuda()
masks = (torch.rand(B * (N + M), generator=g) < 0.97).cuda()
ref_rows = [p * (N + M) for p in range(B)]; src_rows = [p * (N + M) + N for p in range(B)]
f = lambda: ops.superpoint_scores_stack(feats, masks, ref_rows, src_rows, [N] * B, [M] * B, ref_rows, src_rows, True)
out = f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print('superpoint_scores_stack (three kernels): %.1f us per call, sha256 of the scores %s' % (e0.elapsed_time(e1) * 50, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]))
