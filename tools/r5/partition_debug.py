"""Diagnostic (GPU): point_to_node_partition of the 'no overlap (10 m apart)' edge case, HIP against the oracle: are the differences exact ties?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_edge_cases as T
from oracle import se3et_oracle as O
from se3et_amd import ops
from se3et_amd.model import make_cfg
ref, src = T._cases()['no overlap (10 m apart)']
cfg = make_cfg('micro_e'); b = cfg.backbone
pts = torch.from_numpy(np.concatenate([ref, src], 0))
od = O.precompute(pts, torch.tensor([len(ref), len(src)]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
L = od['lengths']
K = cfg.model.num_points_in_patch
for c, name in ((0, 'ref'), (1, 'src')):
    n0 = int(L[-1][:c].sum()); p0 = int(L[1][:c].sum())
    nodes = od['points'][-1][n0:n0 + int(L[-1][c])]; points = od['points'][1][p0:p0 + int(L[1][c])]
    p2n_w, masks_w, knn_w, km_w = O.point_to_node_partition(points, nodes, K)
    p2n_g, masks_g, knn_g, km_g = ops.point_to_node_partition(points.cuda(), nodes.cuda(), K)
    sq = O.pairwise_distance(nodes, points)
    print(name, 'nodes', len(nodes), 'points', len(points), 'K', K, 'point_to_node equal', bool((p2n_g.cpu() == p2n_w).all()))
    kg = knn_g.cpu()
    rows = torch.nonzero((kg != knn_w).any(1))[:, 0].tolist()
    print('  rows differing', len(rows))
    for r in rows[:6]:
        cols = torch.nonzero(kg[r] != knn_w[r])[:, 0].tolist()
        sp = torch.cat((sq[r], torch.tensor([float('inf')])))
        print('   row', r, 'cols', cols[:6], 'oracle', knn_w[r][cols[:6]].tolist(), sp[knn_w[r][cols[:6]]].tolist(), 'hip', kg[r][cols[:6]].tolist(), sp[kg[r][cols[:6]]].tolist())
