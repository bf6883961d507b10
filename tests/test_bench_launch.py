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
    assert line['config']['pairs_sharded'] == 2 * 4 * 8            # every pair of every step owned by exactly one rank
    # the job time is the slowest rank's (rank 1 sleeps 4 ms per step)
    assert line['ms_per_step'] >= 3.9
    assert abs(line['value'] - 2 * 3 * 8 / (line['ms_per_step'] * 3 / 1e3)) <= 0.01 * line['value']


def test_bench_single_rank_and_torchrun_environment():
    rc, out, err = _run(['--gpus', '1'])
    assert rc == 0, err[-2000:]
    assert json.loads(out.strip())['config']['ranks_seen'] == 1
    # under an external launcher (RANK / WORLD_SIZE set) the process is a rank, not a launcher: a mismatching --gpus is refused
    rc, out, err = _run(['--gpus', '2'], {'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1'})
    assert rc != 0 and 'WORLD_SIZE' in err
