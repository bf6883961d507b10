"""The Sinkhorn call of the real forward (8 pairs of the 5k preset; 4 of the KITTI one): its duration and how many patch pairs lie inside the
range of the scaling form (every valid score within 40 of its row's maximum).  python tools/micro/sinkhorn_in_model.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from se3et_amd import functional as SF
from se3et_amd.batched import forward_pairs
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
for variant, preset, pairs in (('se3ete', 'c2_5k', 8), ('se3eti_kitti', 'c3_20k', 4)):
    cfg = make_cfg(variant); b = cfg.backbone
    model = load_synthetic_weights(create_model(cfg)).cuda().eval()
    clouds = []
    for j in range(pairs):
        ref, src, _ = make_pair(preset, index=j); clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda(); lens = torch.tensor([len(c) for c in clouds])
    rec = []
    orig = SF.log_optimal_transport
    def hooked(scores, rm, cm, alpha, iters, inf):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = orig(scores, rm, cm, alpha, iters, inf); e1.record()
        valid = rm[:, :, None] & cm[:, None, :]
        z = torch.where(valid, scores, torch.full_like(scores, float('-inf')))
        m = torch.maximum(z.amax(2), alpha.reshape(1, 1).expand(z.shape[0], z.shape[1]))      # (the dustbin column is part of every row)
        lo = torch.where(valid, scores - m[:, :, None], torch.zeros_like(scores)).amin((1, 2))
        lo = torch.minimum(lo, (alpha - m).amin(1))
        rec.append((e0, e1, tuple(scores.shape), float((lo >= -40).float().mean()), float(lo.min())))
        return out
    SF.log_optimal_transport = hooked
    import se3et_amd.batched as BT, se3et_amd.model as MD
    with torch.no_grad():
        for _ in range(3):
            d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
            d['features'] = torch.ones((pts.shape[0], 1), device='cuda')
            forward_pairs(model, d)
    torch.cuda.synchronize()
    SF.log_optimal_transport = orig
    for e0, e1, shape, frac, lo in rec[-1:]:
        print('%s: Sinkhorn call on %s: %.1f us; %.1f %% of the patch pairs inside the range of the scaling form (widest row: %.1f below its maximum)' % (variant, shape, e0.elapsed_time(e1) * 1e3, 100 * frac, -lo))
