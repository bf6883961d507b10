"""CPU: `python bench.py --gpus 2` launches itself -- two fresh rank processes, gloo rendezvous on 127.0.0.1, round-robin sharding,
barrier + MAX-over-ranks clock, ONE JSON line from rank 0 -- with the GPU step replaced by a host sleep (--fake-device)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--fake-device', '--steps', '3', '--warmup', '1'] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return r.returncode, r.stdout.decode(), r.stderr.decode()


def test_bench_launches_its_own_ranks():
    rc, out, err = _run(['--gpus', '2'])
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['ranks_seen'] == 2
    assert line['config']['pairs_sharded'] == 2 * 4 * 16           # every pair of every step owned by exactly one rank
    # the job time is the slowest rank's (rank 1 sleeps 4 ms per step)
    assert line['ms_per_step'] >= 3.9
    assert abs(line['value'] - 2 * 3 * 16 / (line['ms_per_step'] * 3 / 1e3)) <= 0.01 * line['value']


def test_bench_single_rank_and_torchrun_environment():
    rc, out, err = _run(['--gpus', '1'])
    assert rc == 0, err[-2000:]
    assert json.loads(out.strip())['config']['ranks_seen'] == 1
    # under an external launcher (RANK / WORLD_SIZE set) the process is a rank, not a launcher: a mismatching --gpus is refused
    rc, out, err = _run(['--gpus', '2'], {'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1'})
    assert rc != 0 and 'WORLD_SIZE' in err


def _stub(tmp_path, body):
    path = tmp_path / 'stub_rank.py'
    path.write_text('import json, os, sys, time\nrank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])\n' + body)
    return str(path)


def test_real_launcher_branch_with_stub_ranks(tmp_path, monkeypatch):
    """The launcher itself (not --fake-device): N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* on 127.0.0.1, rank 0's JSON line
    forwarded, the device count taken from the environment / sysfs -- the parent never calls into HIP."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2')
    assert bench.visible_gpus() == 3
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    assert 'torch.cuda' not in open(os.path.join(ROOT, 'bench.py')).read().split('def main():')[0].split('def launch_ranks')[1]
    script = _stub(tmp_path, 'assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == str(rank)\n'
                             'time.sleep(0.2 * rank)\n'
                             'if rank == 0: print(json.dumps({"ranks_seen": world, "argv": sys.argv[1:]}))\n')
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.launch_ranks(3, ['--steps', '2'], script=script)
    assert rc == 0
    line = json.loads(buf.getvalue().strip())
    assert line == {'ranks_seen': 3, 'argv': ['--steps', '2']}


def test_launcher_ends_the_other_ranks_when_one_dies(tmp_path):
    """A rank that dies before the rendezvous must not leave the launcher waiting: the first non-zero exit ends the remaining children
    and becomes the launcher's status."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    script = _stub(tmp_path, 'if rank == 1: sys.exit(7)\ntime.sleep(120)\n')
    t0 = time.monotonic()
    rc = bench.launch_ranks(3, [], script=script)
    assert rc == 7
    assert time.monotonic() - t0 < 30


def test_launcher_gives_every_rank_a_disjoint_slice_of_the_host_cores():
    """a rank's host side is three Python threads launching ~1000 kernels per step each (3.7 busy cores per rank);
    the launcher pins every rank to its own cores // N cores, and the in-flight batches are sized by that budget."""
    sys.path.insert(0, ROOT)
    import bench
    # the arithmetic: equal disjoint slices, remainder unused, nothing when there are fewer cores than ranks
    avail = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13}
    sets = [bench.rank_cpu_set(r, 4, avail) for r in range(4)]
    assert sets == [{3, 4}, {5, 6}, {7, 8}, {9, 10}]
    assert bench.rank_cpu_set(0, 16, {0, 1, 2}) is None
    assert bench.inflight_for_cores(32, None) == 3 and bench.inflight_for_cores(3, None) == 2 and bench.inflight_for_cores(2, None) == 1
    assert bench.inflight_for_cores(8, 3) == 3
    assert bench.inflight_for_cores(1, None, blocking_waits=True) == 3 and bench.inflight_for_cores(2, 4, blocking_waits=True) == 4      # threads that sleep while they wait
    try:
        bench.inflight_for_cores(3, 3)
        raise AssertionError('an --inflight above the core budget must be refused')
    except SystemExit as e:
        assert 'lower --inflight' in str(e)
    # the real launcher: two fake-device ranks report the cores they run on
    have = len(os.sched_getaffinity(0))
    if have < 2:
        return
    rc, out, err = _run(['--gpus', '2'])
    assert rc == 0, err[-2000:]
    cpus = json.loads(out.strip())['config']['rank_cpus']
    assert len(cpus) == 2 and all(len(c) == have // 2 for c in cpus), cpus
    assert not set(cpus[0]) & set(cpus[1]), cpus


def test_train_bench_launches_its_own_ranks_and_keeps_the_replicas_in_sync():
    """tools/train_bench.py's launcher branch at world size 2 over gloo -- the self-launch (bench.launch_ranks), the
    rendezvous, se3et_amd.training.distributed_model (DistributedDataParallel) and make_optimizer (lr x world size) around a stub model on
    the CPU, barrier + MAX-over-ranks clock, ONE JSON line from rank 0.  Every rank feeds its own data: the replicas stay identical only if
    the gradient all-reduce runs."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'train_bench.py'), '--gpus', '2', '--fake-device', '--steps', '3', '--warmup', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd='/tmp')
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.lstrip().startswith('{')]
    assert len(lines) == 1, r.stdout.decode()
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['steps'] == 3
    assert line['replicas_in_sync'] is True
    assert abs(line['lr'] - 2e-2) < 1e-12                      # the reference scales the learning rate by the world size (base_trainer.py:191-196)
