"""Round-6 diagnostic: bench.py's sequence in small -- [blocking-sync request] -> 3 host threads / streams with 8-pair forwards -> 8-pair steps
on the default stream -> one-pair forwards WITH registration -- with the intermediate state printed where the one-pair forward fails."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import se3et_amd
if '--early-sync' in sys.argv:
    print('request:', se3et_amd.request_blocking_sync(0), flush=True)
import numpy as np
import torch

from se3et_amd import _lib, ops
from se3et_amd.batched import forward_pairs
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair

torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
b = cfg.backbone


def inputs(first, pairs):
    clouds = []
    for j in range(pairs):
        ref, src, _ = make_pair('c2_5k', index=first + j)
        clouds += [ref, src]
    return torch.from_numpy(np.concatenate(clouds, 0)).to(dev), torch.tensor([len(c) for c in clouds], dtype=torch.int64)


def forward(pts, lens, **kw):
    data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    data['features'] = torch.ones((pts.shape[0], 1), dtype=torch.float32, device=dev)
    return forward_pairs(model, data, **kw) if len(lens) > 2 else [model(data, **kw)]


big = [inputs(8 * s, 8) for s in range(6)]
if '--threads' in sys.argv:
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]

    def run(t):
        with torch.cuda.stream(streams[t]):
            for s in (t, t + 3):
                forward(*big[s])
            streams[t].synchronize()
    th = [threading.Thread(target=run, args=(t,)) for t in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    print('threads done', flush=True)
if '--timing' in sys.argv:
    ops.KERNEL_TIMINGS = {}
    _lib.lib().se3_debug_kernel_timing(1)
if '--quiet' in sys.argv:
    for s in range(2):
        forward(*big[s])
    torch.cuda.synchronize()
    print('default-stream 8-pair steps done', flush=True)
if '--timing' in sys.argv:
    ops.KERNEL_TIMINGS = None
for i in range(int(os.environ.get("N", 4))):
    pts, lens = inputs(1000 + i, 1)
    try:
        out = forward(pts, lens)[0]
        torch.cuda.synchronize()
        print('1-pair', i, 'ok corr', out['ref_corr_points'].shape[0], 'patches', out['ref_node_corr_indices'].shape[0], flush=True)
    except Exception as e:
        print('1-pair', i, 'FAILED', repr(e)[:200], flush=True)
        out = forward(pts, lens, with_registration=False)[0]
        torch.cuda.synchronize()
        ms = out['matching_scores']
        print('   patches', out['ref_node_corr_indices'].shape[0], 'matching_scores', tuple(ms.shape), 'finite', bool(torch.isfinite(ms).all()),
              'max', float(ms[:, :-1, :-1].max()) if ms.numel() else None,
              'knn masks', int(out['ref_node_corr_knn_masks'].sum()), int(out['src_node_corr_knn_masks'].sum()),
              'feats_f finite', bool(torch.isfinite(out['feats_f']).all()), 'ref_feats_c finite', bool(torch.isfinite(out['ref_feats_c']).all()), flush=True)
