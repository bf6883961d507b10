"""CPU, world_size 2 over gloo: pair sharding and the MAX-over-ranks timing used by bench.py --gpus N."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from se3et_amd import sharding
    r, w, _ = sharding.init_distributed('gloo')
    mine = sharding.shard_pairs(7, r, w)
    elapsed = sharding.max_over_ranks(1.0 + r)              # the slowest rank defines the job time
    total = sharding.sum_over_ranks(len(mine))
    sharding.barrier()
    gathered = sharding.gather_objects({'rank': r, 'pairs': mine})
    q.put((r, mine, elapsed, total, gathered))
    torch.distributed.destroy_process_group()


def test_two_rank_sharding():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(30)
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    assert res[0][2] == 2.0 and res[1][2] == 2.0
    assert res[0][3] == 7.0
    assert sorted(sum((g['pairs'] for g in res[0][4]), [])) == list(range(7))
    assert res[1][4] is None


def test_shard_covers_all_pairs_once():
    from se3et_amd.sharding import shard_pairs
    for world in (1, 2, 4, 8):
        seen = sorted(sum((shard_pairs(64, r, world) for r in range(world)), []))
        assert seen == list(range(64))
