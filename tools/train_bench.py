"""Training-step benchmark (BASELINE.json configs[4]): SE3ET-E forward + backward + Adam on synthetic 5k+5k pairs, one pair per rank
and step, DistributedDataParallel over RCCL when --gpus > 1 (self-launching like bench.py).
    python tools/train_bench.py --gpus 1 --steps 5 --warmup 2 [--pair c2_5k] [--variant se3ete]
Prints one JSON line: seconds per step (MAX over ranks), pairs/s over all ranks, the loss of the last step."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_fake(args):
    """Everything of the multi-process training contract except the GPU and the model: a stub network behind the SAME wrapper, optimizer
    factory, barrier and clock as the real run; rank r feeds its own data, the gradient all-reduce keeps the replicas identical."""
    from types import SimpleNamespace
    from se3et_amd import sharding
    from se3et_amd.training import distributed_model, make_optimizer
    rank, world, _ = sharding.init_distributed('gloo')
    if world != args.gpus:
        raise SystemExit('train_bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    torch.manual_seed(0)                                      # the same initial replica on every rank
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    net = distributed_model(model) if world > 1 else model
    opt = make_optimizer(net, SimpleNamespace(optim=SimpleNamespace(lr=1e-2, weight_decay=0.0)), world)
    g = torch.Generator().manual_seed(100 + rank)             # ... and different data per rank

    def step():
        x = torch.randn(32, 8, generator=g)
        loss = (net(x) - x.sum(1, keepdim=True)).pow(2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(args.warmup):
        step()
    sharding.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sharding.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    spread = sharding.max_over_ranks(float(flat.double().sum())) + sharding.max_over_ranks(-float(flat.double().sum()))     # max - min over ranks
    if rank == 0:
        print(json.dumps({'metric': 'training launcher self-test (no GPU work)', 's_per_step': round(dt / max(args.steps, 1), 5), 'n_gpus': world,
                          'ranks_seen': torch.distributed.get_world_size() if world > 1 else 1, 'steps': args.steps, 'warmup': args.warmup,
                          'loss': float(loss.detach()), 'replicas_in_sync': abs(spread) < 1e-9, 'lr': opt.param_groups[0]['lr'],
                          'data': 'none (fake device)'}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--variant', default='se3ete')
    ap.add_argument('--pair', default='c2_5k')
    ap.add_argument('--profile', action='store_true', help='GPU time of the backward of each HIP op (HIP events), one extra step')
    ap.add_argument('--fake-device', action='store_true', help='launcher self-test (no GPU): the self-launch, the gloo rendezvous, DistributedDataParallel '
                    'through se3et_amd.training.distributed_model with a stub model on the CPU, the barrier + MAX-over-ranks clock and the one JSON '
                    'line run for real (tests/test_bench_launch.py)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import bench
        raise SystemExit(bench.launch_ranks(args.gpus, sys.argv[1:], script=__file__))
    if args.fake_device:
        return run_fake(args)
    import se3et_amd
    if int(os.environ.get('WORLD_SIZE', '1')) == 1 or os.environ.get('SE3_BLOCKING_SYNC') == 'force':      # (as bench.py: measured with one rank only)
        se3et_amd.request_blocking_sync(int(os.environ.get('LOCAL_RANK', '0')))  # the process's first GPU call (se3et_amd/__init__.py)
    from se3et_amd import sharding
    from se3et_amd.data import registration_collate_fn_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    from se3et_amd.training import OverallLoss, distributed_model, make_optimizer, train_step
    rank, world, local = sharding.init_distributed('nccl')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cfg = make_cfg(args.variant)
    model = load_synthetic_weights(create_model(cfg)).to(dev).train()
    net = distributed_model(model, dev) if world > 1 else model
    loss_fn, opt = OverallLoss(cfg), make_optimizer(net, cfg, world)
    b = cfg.backbone
    rng = np.random.RandomState(rank)
    batches = []
    for s in range(args.steps + args.warmup):
        ref, src, T = make_pair(args.pair, index=s * world + rank)
        d = dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32), src_feats=np.ones((len(src), 1), np.float32),
                 transform=T)
        batches.append(d)
    def step(i):
        dd = registration_collate_fn_stack_mode([batches[i]], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits, device=dev)
        return train_step(net, dd, loss_fn, opt, rng=rng)[0]
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    sharding.barrier(dev)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        losses = step(i)
    torch.cuda.synchronize()
    sharding.barrier(dev)
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps({'metric': 'training step (fwd + bwd + Adam), %s on %s pairs' % (args.variant, args.pair), 's_per_step': round(dt / args.steps, 4),
                          'pairs_per_s': round(world * args.steps / dt, 3), 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'loss': float(losses['loss'].detach()), 'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                          'backward': 'HIP kernels for KPConv / GroupNorm / LayerNorm / Sinkhorn / embedding / max-pool / row gather; attention (RPE self, plain cross, equivariant cross): hand-derived backward, logits recomputed by the HIP kernel, batched GEMMs (se3et_amd/attention_bwd.py)', 'data': 'synthetic'}), flush=True)
    if args.profile and world > 1 and rank == 0:
        print('--profile is a single-GPU option (a forward / backward through the DDP wrapper on one rank would wait for its peers): skipped')
    if args.profile and world == 1:
        from se3et_amd import autograd as AG
        AG.BACKWARD_TIMINGS = {}
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        dd = registration_collate_fn_stack_mode([batches[0]], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits, device=dev)
        opt.zero_grad(set_to_none=True)
        e0.record()
        out = model(dd, train=True, rng=rng)
        loss = loss_fn(out, dd)['loss']
        e1.record()
        loss.backward()
        e2.record()
        torch.cuda.synchronize()
        print('forward + loss %.1f ms, backward %.1f ms' % (e0.elapsed_time(e1), e1.elapsed_time(e2)))
        rows = sorted(((sum(a.elapsed_time(bb) for a, bb in v), len(v), k) for k, v in AG.BACKWARD_TIMINGS.items()), reverse=True)
        for ms, n, k in rows:
            print('  backward of %-24s x%-3d %8.1f ms' % (k, n, ms))
        AG.BACKWARD_TIMINGS = None
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
