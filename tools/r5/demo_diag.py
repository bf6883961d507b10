"""Diagnostic (GPU): the HIP path on the reference's real pair (tests/golden/demo_se3ete.npz): tables against the reference's checksums,
features with the index-ordered tie rule and with the reference's choice patched into the rows whose neighbour SET differs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import index_checksum, tie_canonical, rel_err
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg

g = np.load(os.path.join(ROOT, 'tests/golden/demo_se3ete.npz'))
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg), int(g['synth_seed'])).cuda().eval()
ref, src = g['ref'], g['src']
pts = torch.from_numpy(np.concatenate([ref, src], 0)).cuda()
b = cfg.backbone
def pyramid():
    dd = precompute_data_stack_mode(pts, torch.tensor([len(ref), len(src)]), b.num_stages, b.init_voxel_size, b.init_radius, [38, 36, 36, 38])
    dd['features'] = torch.ones((pts.shape[0], 1), device='cuda')
    return dd
dd = pyramid()
print('lengths', [l.tolist() for l in dd['lengths']], 'want', g['lengths'].tolist())
P = [p.cpu().numpy() for p in dd['points']]
geo = {}
for i in range(4): geo['neighbors', i] = (P[i], P[i])
for i in range(3): geo['subsampling', i] = (P[i + 1], P[i]); geo['upsampling', i] = (P[i], P[i + 1])
for (key, i), (q, s) in geo.items():
    t = dd[key][i].cpu().numpy()
    c = tie_canonical(q, s, t)
    print(key, i, t.shape, 'exact', index_checksum(t) == int(g['checksum/' + key][i]), 'rowset', index_checksum(np.sort(t, 1)) == int(g['rowset/' + key][i]),
          'tiecanon', index_checksum(c[0]) == int(g['tiecanon/' + key][i]), 'tierows', c[1], int(g['tierows/' + key][i]))
print('points_last', np.array_equal(P[-1], g['points_last']))

def run(dd, label):
    taps = {}
    model.transformer.transformer.layer_tap = lambda i, t: taps.__setitem__(i, t)
    out = model(dd)
    rs = int(g['row_step'])
    print(label, 'feats_c', rel_err(out['feats_c'][::rs, :, ::4], g['p0/feats_c']), 'feats_f', rel_err(out['feats_f'][::4 * rs], g['p0/feats_f']))
    for i in range(len(g['blocks'])):
        print(label, 'layer', i, g['blocks'][i], rel_err(taps[i][..., ::rs, :], g['op/layer_%d/out0' % i]))
    for k in ('ref_feats_c', 'src_feats_c', 'estimated_transform'):
        print(label, k, rel_err(out[k], g['p0/' + k]))
    gp = set(zip(out['ref_node_corr_indices'].tolist(), out['src_node_corr_indices'].tolist()))
    wp = set(zip(g['p0/ref_node_corr_indices'].tolist(), g['p0/src_node_corr_indices'].tolist()))
    print(label, 'superpoint pairs', len(gp), len(wp), 'common', len(gp & wp), 'num_corr', out['ref_corr_points'].shape[0], int(g['p0/num_corr']))
    return out

run(dd, 'index-ordered ties:')
dd2 = pyramid()
for key in ('neighbors', 'subsampling', 'upsampling'):
    for i in range(len(dd2[key])):
        rows = torch.from_numpy(g['patch/%s_%d_rows' % (key, i)]).long().cuda()
        if rows.numel():
            dd2[key][i][rows] = torch.from_numpy(g['patch/%s_%d_vals' % (key, i)]).long().cuda()
        t = dd2[key][i].cpu().numpy()
        print('patched', key, i, len(rows), 'rowset', index_checksum(np.sort(t, 1)) == int(g['rowset/' + key][i]))
run(dd2, 'reference tie choice :')
