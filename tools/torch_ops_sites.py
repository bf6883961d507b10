"""Which call sites of the 8-pair forward launch torch's own small kernels (copies, cats, fills, element-wise)?  A TorchDispatchMode records
every aten op with the innermost se3et_amd frame that issued it; GPU time per op from a second, profiled run (self device time by op name and
input sizes, spread over the sites by call count).  python tools/torch_ops_sites.py [pairs]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from se3et_amd.batched import forward_pairs
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda'); cfg = make_cfg('se3ete'); b = cfg.backbone
model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
clouds = []
for j in range(pairs):
    ref, src, _ = make_pair('c2_5k', index=j); clouds += [ref, src]
pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev); lens = torch.tensor([len(c) for c in clouds])
feats = torch.ones((pts.shape[0], 1), device=dev)
def step():
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = feats
    return forward_pairs(model, d) if pairs > 1 else model(d)

FILLS = ('aten::zeros', 'aten::zeros_like', 'aten::fill_', 'aten::zero_', 'aten::ones', 'aten::ones_like', 'aten::full', 'aten::new_zeros', 'aten::copy_', 'aten::_to_copy', 'aten::clone', 'aten::cat', 'aten::contiguous')
SKIP = ('aten::view', 'aten::_unsafe_view', 'aten::reshape', 'aten::t', 'aten::transpose', 'aten::permute', 'aten::expand', 'aten::slice',
        'aten::select', 'aten::unsqueeze', 'aten::squeeze', 'aten::detach', 'aten::alias', 'aten::as_strided', 'aten::empty', 'aten::split',
        'aten::unbind', 'aten::_local_scalar_dense', 'aten::empty_like', 'aten::empty_strided', 'aten::new_empty', 'aten::unfold', 'aten::chunk',
        'aten::narrow', 'aten::lift_fresh', 'aten::is_nonzero', 'aten::item')

class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.count = collections.Counter(); self.numel = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = 'aten::' + func.__name__.split('.')[0]
        if name not in SKIP:
            f, chain = sys._getframe(1), []
            while f is not None and len(chain) < 3:
                fn = f.f_code.co_filename
                if 'se3et_amd' in fn and 'torch' not in fn.split('se3et_amd')[-1]:
                    chain.append('%s:%d' % (fn.split('se3et_amd/')[-1], f.f_lineno))
                f = f.f_back
            site = ' < '.join(chain) if chain else '?' 
            t = out if torch.is_tensor(out) else (out[0] if isinstance(out, (tuple, list)) and out and torch.is_tensor(out[0]) else None)
            on_gpu = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + ([t] if t is not None else []))
            if on_gpu:
                self.count[(name, site)] += 1
                self.numel[(name, site)] += t.numel() if t is not None else 0
        return out

with torch.no_grad():
    for _ in range(2): step()
    torch.cuda.synchronize()
    with Sites() as S:
        step()
    torch.cuda.synchronize()
per_site = collections.defaultdict(lambda: [0, 0])
for (name, site), n in S.count.items():
    per_site[site][0] += n; per_site[site][1] += S.numel[(name, site)]
print('aten ops on GPU tensors in one forward of %d pairs: %d at %d sites (views / allocations not counted)' % (pairs, sum(S.count.values()), len(per_site)))
for (name, site), n in sorted(S.count.items(), key=lambda kv: -S.numel[kv[0]])[:60]:
    print('x%-3d %-26s %12d elements  %s' % (n, name, S.numel[(name, site)], site))
print('--- fills / copies by site')
for (name, site), n in sorted(((k, v) for k, v in S.count.items() if k[0] in FILLS), key=lambda kv: -kv[1])[:50]:
    print('x%-3d %-26s %12d elements  %s' % (n, name, S.numel[(name, site)], site))
print('--- launches by site (all ops)')
for site, (n, el) in sorted(per_site.items(), key=lambda kv: -kv[1][0])[:40]:
    print('x%-3d %12d elements  %s' % (n, el, site))
