"""Several batches in flight (`bench.py --inflight N`: one host thread + one HIP stream each) against the same batches run one after the
other.  The model is FRESH (cold weight-piece, index-table and composed-projection caches): the first use is where a cache shared by the
streams can be read on one stream before the kernels that fill it have run on another (se3et_amd/ops.py: _Shared)."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(cfg, first_pair, pairs):
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.synthetic import make_pair
    b = cfg.backbone
    clouds = []
    for j in range(pairs):
        ref, src, _ = make_pair('c2_5k', index=first_pair + j)
        clouds += [ref, src]
    pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
    lens = torch.tensor([len(c) for c in clouds])

    def run():
        data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
        data['features'] = torch.ones((pts.shape[0], 1), device='cuda')
        return data
    return run


@pytest.mark.parametrize('threads,schedule', [(2, 'throughput'), (3, 'throughput'), (3, 'roofline'), (3, 'exclusive')])
def test_batches_in_flight_give_the_sequential_results(threads, schedule):
    """... under every schedule of se3et_amd.batched.set_schedule (round 6: 'roofline' switches the backbone between two chains where it
    announces its pyramid stage, one of them shared with the transformer sections)."""
    from se3et_amd import batched, ops
    from se3et_amd.batched import forward_pairs
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    batched.set_schedule(schedule)
    cfg = make_cfg('se3ete')
    model = load_synthetic_weights(create_model(cfg)).cuda().eval()
    ops.clear_weight_caches()
    builders = [_batch(cfg, 2 * t, 2) for t in range(threads)]
    streams = [torch.cuda.Stream() for _ in range(threads)]
    got, failed = [None] * threads, []
    gate = threading.Barrier(threads)

    def work(t):
        try:
            with torch.cuda.stream(streams[t]), torch.no_grad():
                gate.wait()
                outs = forward_pairs(model, builders[t]())
                got[t] = [{k: v.detach().clone() for k, v in o.items() if torch.is_tensor(v)} for o in outs]
                streams[t].synchronize()
        except BaseException as e:       # (a thread's exception would otherwise only be printed)
            failed.append(e)

    pool = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    if failed:
        raise failed[0]
    torch.cuda.synchronize()
    with torch.no_grad():
        for t in range(threads):
            want = forward_pairs(model, builders[t]())
            for p, (g, w) in enumerate(zip(got[t], want)):
                assert g['ref_corr_points'].shape == w['ref_corr_points'].shape, (t, p)
                assert torch.equal(g['ref_node_corr_indices'], w['ref_node_corr_indices']), (t, p)
                for key in ('ref_feats_c', 'src_feats_c', 'estimated_transform'):
                    assert float((g[key] - w[key]).abs().max()) <= 1e-5 * max(1.0, float(w[key].abs().max())), (t, p, key)
    batched.set_schedule('throughput')


def test_blocking_sync_is_the_first_gpu_call_or_nothing():
    """ADVICE round 5 + the round-6 measurement (profiles/r06_blocking_sync.txt): the blocking-wait flag works only as the process's FIRST GPU
    call, for the device the rank will use.  (a) a fresh interpreter: request, then torch work -- the process CPU time spent in a wait is a
    small part of the wall time; (b) once torch has initialised the runtime the request is refused and changes nothing (set late it is
    either ineffective or breaks the waits of streams that already exist).  Importing the package alone touches no GPU."""
    import json
    import os
    import subprocess
    import sys
    code = '''
import json, sys, time
import se3et_amd
early = "%s" == "early"
status = se3et_amd.request_blocking_sync(0) if early else None
import torch
torch.cuda.set_device(0)
x = torch.randn(8192, 8192, device="cuda")
torch.cuda.synchronize()
if not early:
    status = se3et_amd.request_blocking_sync(0)
def wait_cost():
    y = x
    for _ in range(40):
        y = (y @ x) * 1e-4
    c0, w0 = time.process_time(), time.perf_counter()
    torch.cuda.synchronize()
    return time.process_time() - c0, time.perf_counter() - w0
wait_cost()
print(json.dumps({"status": status, "wait": wait_cost(), "reported": se3et_amd.blocking_sync_status(0)}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    env.pop('SE3_BLOCKING_SYNC', None)
    out = {}
    for when in ('early', 'late'):
        r = subprocess.run([sys.executable, '-c', code % when], capture_output=True, text=True, env=env, timeout=180)
        assert r.returncode == 0, r.stderr[-2000:]
        out[when] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['early']['status'] == 'set' and out['early']['reported'] == 'set', out
    cpu, wall = out['early']['wait']
    assert wall > 0.05 and cpu < 0.35 * wall, out                    # there was something to wait for, and the waiting thread slept
    assert out['late']['status'].startswith('too late'), out
    cpu_l, wall_l = out['late']['wait']
    assert abs(wall_l - wall) < 0.5 * wall, out                      # same wall time either way
