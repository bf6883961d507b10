import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import make_cfg
from se3et_amd.synthetic import make_pair
for variant, preset, pairs in (('se3ete', 'c2_5k', 8), ('se3eti_kitti', 'c3_20k', 4)):
    cfg = make_cfg(variant); b = cfg.backbone
    clouds = []
    for j in range(pairs):
        ref, src, _ = make_pair(preset, index=j); clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda(); lens = torch.tensor([len(c) for c in clouds])
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for i, t in enumerate(d[key]):
            n_support = d['points'][i + (1 if key == 'upsampling' else 0)].shape[0] if key != 'neighbors' else d['points'][i].shape[0]
            real = ((t >= 0) & (t < n_support)).float().mean().item()
            print('%s %s stage %d: table %s, real entries %.1f %%' % (variant, key, i, tuple(t.shape), 100 * real))
