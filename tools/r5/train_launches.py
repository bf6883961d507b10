"""Where the launches of one training step come from (the step is launch bound: ~3 700 launches in ~50 ms).  torch.profiler over ONE step
after two warm-up steps: every GPU kernel is attributed to its outermost CPU range -- a module of depth <= 2 in the forward (ranges pushed by
hooks), an autograd node in the backward, the optimizer -- and, below it, to the aten / custom op that launched it.
    python tools/r5/train_launches.py [top]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity, record_function
from se3et_amd.data import registration_collate_fn_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
from se3et_amd.training import OverallLoss, make_optimizer
from se3et_amd import autograd as AG
AG.PROFILE_RANGES = True

top = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda')
cfg = make_cfg('se3ete'); b = cfg.backbone
model = load_synthetic_weights(create_model(cfg)).to(dev).train()
loss_fn, opt = OverallLoss(cfg), make_optimizer(model, cfg, 1)
rng = np.random.RandomState(0)
ranges = {}
for name, m in model.named_modules():
    if name and name.count('.') <= 1:
        def pre(mod, args, name=name):
            r = record_function('fwd:' + name); r.__enter__(); ranges.setdefault(id(mod), []).append(r)
        def post(mod, args, out):
            ranges[id(mod)].pop().__exit__(None, None, None)
        m.register_forward_pre_hook(pre); m.register_forward_hook(post)


def step(i, prof=False):
    ref, src, T = make_pair('c2_5k', index=i)
    d = dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32), src_feats=np.ones((len(src), 1), np.float32), transform=T)
    with record_function('collate'):
        dd = registration_collate_fn_stack_mode([d], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits, device=dev)
    out = model(dd, train=True, rng=rng)
    with record_function('loss'):
        losses = loss_fn(out, dd)
    opt.zero_grad(set_to_none=True)
    losses['loss'].backward()
    with record_function('optimizer'):
        opt.step()


for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(2)
    torch.cuda.synchronize()
by_root = collections.Counter(); by_root_op = collections.defaultdict(collections.Counter); by_kernel = collections.defaultdict(collections.Counter)
total = 0
for ev in prof.events():
    ks = [k for k in getattr(ev, 'kernels', [])]
    if not ks:
        continue
    chain = [ev]
    while chain[-1].cpu_parent is not None:
        chain.append(chain[-1].cpu_parent)
    names = [e.name for e in chain]
    # outermost informative range: forward module of depth <= 2 (innermost such), autograd node, optimizer / loss / collate
    root = next((n for n in names if n.startswith('fwd:')), None)
    node = next((n for n in reversed(names) if n.startswith('autograd::engine::evaluate_function')), None)
    if node is not None:
        root = 'bwd:' + node.split(': ', 1)[1]
        hip = next((n for n in names if n.startswith('hipbwd:')), None)
        if hip is not None:
            root = hip
    if root is None:
        root = next((n for n in reversed(names) if n in ('optimizer', 'loss', 'collate')), names[-1])
    op = names[0]
    by_root[root] += len(ks); by_root_op[root][op] += len(ks); total += len(ks)
    for k in ks:
        by_kernel[root][k.name.split('(')[0][-60:]] += 1
print('launches in one step: %d' % total)
for root, n in by_root.most_common(top):
    ops = ', '.join('%s x%d' % (o.replace('aten::', ''), c) for o, c in by_root_op[root].most_common(8))
    print('%5d  %-58s %s' % (n, root[:58], ops))
