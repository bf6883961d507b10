"""bench.py as the driver runs it, on one GPU: (a) `--force-dist` makes the rank initialise torch.distributed on the `nccl` (= RCCL) backend
at world size 1, so bench's OWN collectives -- the device barrier in front of and behind the timed region and the MAX all-reduce of the job
clock on a device tensor -- execute on the GPU (they had never run in any test, `init_distributed` returns early at world
1); (b) the driver's exact step counts (`--steps 20 --warmup 5` is what BENCH_rNN records) run with three batches in flight and every
in-flight stream primed before t0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.lstrip().startswith('{')]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0])


LIGHT = ['--no-cpu-baseline', '--single-pair-steps', '0', '--train-steps', '0', '--roofline-quiet-steps', '0']


def test_bench_collectives_run_on_rccl_at_world_size_one():
    line = _bench(['--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '2', '--force-dist'] + LIGHT)
    cfg = line['config']
    # both come from torch.distributed AFTER the device barrier + MAX all-reduce of the timed region ran through it
    assert cfg['ranks_seen'] == 1 and cfg['collectives'] == 'rccl'
    assert line['n_gpus'] == 1 and line['steps'] == 2 and line['value'] > 0
    # without the flag one rank uses no collective at all
    line = _bench(['--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '2'] + LIGHT)
    assert line['config']['collectives'].startswith('none')


def test_bench_at_the_drivers_step_counts():
    line = _bench(['--gpus', '1', '--steps', '20', '--warmup', '5'] + LIGHT)
    assert line['steps'] == 20 and line['warmup'] == 5 and line['config']['batches_in_flight_per_gpu'] == 3
    assert line['config']['pairs_per_forward'] == 16            # the default since round 6 (profiles/r06_batch_inflight_sweep.txt)
    assert abs(line['value'] - 20 * 16 / (line['ms_per_step'] * 20 / 1e3)) <= 0.01 * line['value']
    assert line['roofline']['launches'] == 20 * 5            # five RPE self-attention calls per SE3ET-E forward, all of them timed
    assert line['host_cpu_s_per_step'] > 0
    # round 5: whole-step HBM figure from the committed counter passes, the dense + GroupNorm family's own roofline, traffic parsed at run time
    rs, rd = line['roofline_step'], line['roofline_dense']
    assert rs['bytes_per_step'] > 20e9 and 0.05 < rs['frac'] < 1.0 and 'profiles/r06_pmc_step.txt' in rs['source']
    assert abs(rs['achieved'] - rs['bytes_per_step'] / (line['ms_per_step'] * 1e-3) / 1e9) <= 0.01 * rs['achieved']
    assert rd['bound'] == 'hbm' and rd['launches'] >= 20 * 30 and 0.0 < rd['frac'] < 1.0
    assert 'profiles/r0' in line['roofline']['traffic_source'] and 'sha256' in line['roofline']['traffic_source']
    assert set(line['roofline']['traffic_per_algorithmic_byte']) == {'eq', 'inv'}
    # host threads sleep while they wait for the GPU (hipDeviceScheduleBlockingSync requested for THIS rank's device after set_device): well
    # under one core per thread
    assert line['config']['host_waits'].endswith('set (device 0)'), line['config']['host_waits']
    # round 6: the same K steps twice more between the same bracket; all three rates with their median / min / max
    # round 6: the default schedule keeps the coarsest backbone stage off the GPU while another batch's transformer runs (the RPE kernels hold
    # their share of HBM in the timed region); the other schedule -- everything overlaps -- is reported from one more region of the same steps
    o = line['other_schedule']
    assert line['config']['schedule'] == 'roofline' and o['schedule'] == 'throughput'
    assert line['roofline']['frac'] > o['roofline_frac'] and line['roofline']['frac'] >= 0.38, (line['roofline']['frac'], o)
    assert o['value'] > 0.95 * line['value']
    d = line['dispersion']
    assert d['regions'] == 3 and len(d['values']) == 3 and abs(d['values'][0] - line['value']) <= 0.01 * line['value']
    assert d['min'] <= d['median'] <= d['max'] and d['spread_rel'] < 0.5
    # (round 6: 1.7-1.9 busy cores per rank with the flag against 3.8-4.2 without; round 5 read 0.8 -- the difference sits in the package, not in
    #  bench.py or the tie pass, and was not found: DESIGN section 5)
    assert line['host_cpu_s_per_step'] < 0.8 * 3 * line['ms_per_step'] * 1e-3
