"""Data-parallel sharding of registration pairs over ranks (one process per GPU).

A pair is an indivisible unit (GroupNorm statistics and cross attention couple ref and src) and
pairs are independent, so the forward path needs NO collective: rank r of W takes pairs r, r+W, r+2W, ...  The only
communication is a final MAX-reduction of the elapsed time (and an optional gather of per-rank results) through
torch.distributed (backend 'nccl' = RCCL on ROCm; 'gloo' in the CPU tests)."""
import os

import torch
import torch.distributed as dist


def rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def init_distributed(backend='nccl', force=False):
    """Initialise torch.distributed from the torchrun environment if WORLD_SIZE > 1.  Returns (rank, world, local_rank).
    force=True (or SE3ET_FORCE_DIST=1): also with ONE rank, so that the collectives of a run (device barrier, MAX all-reduce on a device
    tensor over RCCL) execute on a single GPU -- the only way to exercise them where no multi-GPU node is at hand."""
    rank, world, local = rank_world()
    force = force or os.environ.get('SE3ET_FORCE_DIST', '0') == '1'
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_pairs(num_pairs, rank, world):
    """Indices of the pairs rank `rank` owns (round-robin, as the 64-pair / 8-GPU configuration C4)."""
    if not (0 <= rank < world):
        raise ValueError('rank %d outside world %d' % (rank, world))
    return list(range(rank, num_pairs, world))


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and device.type == 'cuda':
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the slowest rank defines the job time)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_objects(obj):
    """All ranks' python objects on rank 0 (timings / metrics); None elsewhere."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out
