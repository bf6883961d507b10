#!/usr/bin/env python
"""Benchmark of the SE3ET hot path on MI355X:  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): SE3ET-E forward on synthetic 5k+5k point-cloud pairs (3DMatch-sized), fp32,
name-keyed synthetic weights.  A step = one batch of `--batch` (default 16 since round 6: +5 % over 8,
profiles/r06_batch_inflight_sweep.txt) registration pairs per rank through ONE forward
(se3et_amd.batched: clouds stacked ref0, src0, ref1, ...; per pair the results of the single-pair forward): on-GPU stage
pyramid (grid subsampling + 10 radius searches) -> E2PN backbone -> geometric transformer -> superpoint matching ->
Sinkhorn -> local-to-global registration.  `--batch 1` runs the reference-shaped single-pair forward.  By default three such
batches are in flight per GPU (`--inflight`: one host thread + HIP stream each; a batch's host synchronisations and under-filled
kernels are covered by the other batches).  The raw pairs are resident in HBM before the timed region.  Pairs are sharded over ranks with no collective on the data path (weak scaling);
the job time is the MAX over ranks; value = pairs / second over all ranks.

Prints ONE JSON line: pairs/s plus a `roofline` object for the RPE self-attention kernels (per-launch HIP events on the
launch stream, live over the timed region; `roofline.quiet`: the same kernels with one batch in flight), a `roofline_kpconv` object (the
fused KPConv kernel -- the largest family by time -- against the f16 matrix-core peak, event pairs around its launches), a `cpu_baseline` object (the CPU oracle timed on this box's host cores) and a
`single_pair` object (the reference-shaped one-pair-per-forward path: pairs/s, kernel launches, host synchronisations).

Multi-GPU: one process per GPU.  Under `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) every process
is one rank.  Started directly with `--gpus N > 1`, this process only LAUNCHES: before anything touches a GPU it starts N fresh
child processes of itself (the reference's one-process-per-GPU launch, geotransformer/engine/base_trainer.py:66-78), one rank
each, rendezvous on 127.0.0.1, prints rank 0's JSON line and exits with the worst child status.  `--gpus 8 --batch 8` is
BASELINE.json configs[3] (64 independent pairs per step, 8 per rank)."""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HBM traffic of one RPE self-attention call relative to its algorithmic bytes: PARSED AT RUN TIME from the committed rocprofv3 PMC passes
# (profiles/rNN_pmc_attention_raw.txt: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of tools/pmc_attention.py on the stack-mode kernels
# at the bench shape, 16 clouds per launch, collected by tools/pmc_passes.sh; one line per (kernel, grid, counter): average KiB per dispatch;
# '#' lines: the collecting commit and the algorithmic bytes of the two calls).  Counters cannot be read inside this process, so the ratio
# of that run is applied to the calls of this run.  gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for the
# 16-B/lane streaming reads of rpe_bias_kernel and the operand split (x6_split_kernel; attn_split_kv_kernel in files of earlier commits); the attention kernel's counters as reported.
PMC_TRAFFIC_FILES = ('profiles/r06_pmc_attention_raw.txt', 'profiles/r05_pmc_attention_raw.txt', 'profiles/r04_pmc_attention_raw.txt')
PMC_ALGORITHMIC_MB_R04 = {'eq': 2549.0, 'inv': 2222.9}      # (the round-4 file carries no '# algorithmic' lines: tools/pmc_attention.py at its shape)


def load_pmc_traffic():
    """-> (ratios {'eq', 'inv'}: HBM-side bytes per algorithmic byte of one call, source string) from the newest committed PMC file."""
    import hashlib
    for rel in PMC_TRAFFIC_FILES:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        text = open(path).read()
        algo, commit = dict(PMC_ALGORITHMIC_MB_R04) if 'r04' in rel else {}, 'not recorded in the file'
        kib = {'eq': 0.0, 'inv': 0.0}
        kind = None
        for line in text.splitlines():
            line = line.strip()
            if line.startswith('#'):
                if 'commit' in line:
                    commit = line.split('commit', 1)[1].strip(' :')
                if 'algorithmic bytes per call' in line:          # "# A=6 eq=1: algorithmic bytes per call N"
                    algo['eq' if 'eq=1' in line else 'inv'] = float(line.rsplit(' ', 1)[1]) / 1e6
                continue
            parts = line.rsplit(',', 4)                            # "name<...>",grid,counter,count,KiB   (the name holds commas)
            if len(parts) != 5:
                continue
            name, _, counter, _, value = parts
            if 'rpe_bias_kernel' in name:                          # template argument 4: the equivariant term
                args_t = name.split('<', 1)[1].split(',')
                kind = 'eq' if args_t[3].strip() == 'true' else 'inv'
            if kind is None:
                continue
            streaming = 'rpe_bias_kernel' in name or 'attn_split_kv_kernel' in name or 'x6_split_kernel' in name
            kib[kind] += float(value) * (2.0 if (counter == 'FETCH_SIZE' and streaming) else 1.0)
        if kib['eq'] > 0 and kib['inv'] > 0 and 'eq' in algo and 'inv' in algo:
            ratios = {k: kib[k] * 1024.0 / 1e6 / algo[k] for k in kib}
            return ratios, '%s (collected at commit %s; sha256 %s)' % (rel, commit, hashlib.sha256(text.encode()).hexdigest()[:12])
    return None, 'no PMC file under profiles/'


def load_pmc_step():
    """HBM-side bytes of ONE 8-pair bench step by kernel family from the committed whole-step PMC passes (tools/pmc_step.sh ->
    profiles/r06_pmc_step.txt) -> (total bytes, {family: bytes}, source) or (None, None, reason)."""
    import hashlib
    rel = 'profiles/r06_pmc_step.txt' if os.path.exists(os.path.join(ROOT, 'profiles/r06_pmc_step.txt')) else 'profiles/r05_pmc_step.txt'
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, None, rel + ' missing'
    text = open(path).read()
    fams, total, commit = {}, None, 'not recorded in the file'
    for line in text.splitlines():
        if line.startswith('#'):
            if 'commit' in line:
                commit = line.split('commit', 1)[1].strip(' :')
            continue
        parts = line.strip().rsplit(',', 4)
        if len(parts) != 5:
            continue
        name = parts[0].strip('"')
        nbytes = (float(parts[4]) + float(parts[3])) * 1024.0       # FETCH (corrected) + WRITE
        if name == 'total':
            total = nbytes
        else:
            fams[name] = nbytes
    if total is None:
        return None, None, rel + ' holds no total line'
    return total, fams, '%s (collected at commit %s; sha256 %s)' % (rel, commit, hashlib.sha256(text.encode()).hexdigest()[:12])


HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s is achievable


def visible_gpus():
    """GPUs this process may use WITHOUT touching the HIP runtime (the launcher parent must never initialise a GPU: its children do):
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES if set, else the KFD topology (nodes with SIMDs).  None when neither is readable."""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(',') if t.strip() != ''])
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        count = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            count += int(props.get('simd_count', '0')) > 0
        return count
    except (OSError, ValueError):
        return None


def rank_cpu_set(local_rank, local_world, available=None):
    """The host cores of one rank: the `local_rank`-th of `local_world` equal, disjoint slices of the cores this process may run on
    (sorted; the remainder stays unused).  A rank's host side is `--inflight` Python threads that launch ~1000 kernels per step each --
    3.7 busy cores per rank at the default 3 batches in flight (BENCH_r04 host_cpu_s_per_step) -- so eight ranks must not wander over each
    other's cores.  None when the cores cannot be split (fewer cores than ranks, or no affinity interface)."""
    if available is None:
        try:
            available = os.sched_getaffinity(0)
        except (AttributeError, OSError):
            return None
    cores = sorted(available)
    per = len(cores) // max(1, local_world)
    if per < 1:
        return None
    return set(cores[local_rank * per:(local_rank + 1) * per])


def thread_cpu_times():
    """{tid: (name, user + system CPU seconds)} of every thread of this process (Linux /proc)."""
    out, tick = {}, os.sysconf('SC_CLK_TCK')
    for tid in os.listdir('/proc/self/task'):
        try:
            stat = open('/proc/self/task/%s/stat' % tid).read()
            name = stat[stat.index('(') + 1:stat.rindex(')')]
            f = stat[stat.rindex(')') + 2:].split()
            out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tick)
        except (OSError, ValueError):
            pass
    return out


def inflight_for_cores(cores_per_rank, requested=None, batch=8, blocking_waits=False):
    """Batches in flight per rank that `cores_per_rank` host cores carry.  With SPINNING waits (the runtime's default) every in-flight thread
    is a busy core while it waits for the GPU: one core per thread plus one for the process.  With BLOCKING waits (se3et_amd asks for them at
    import) a thread sleeps while it waits and the three threads of a rank keep 0.7 cores busy (host_cpu_s_per_step 0.012 s per 16.7 ms
    step): three in flight on any slice, two threads per core beyond that.  Default 3 (same-box optimum on an unshared host); a request
    above the budget is refused -- lower --inflight."""
    if batch <= 1:
        return 1 if requested is None else requested
    if not cores_per_rank:
        budget = 3
    elif blocking_waits:
        budget = max(3, 2 * cores_per_rank)
    else:
        budget = max(1, cores_per_rank - 1)
    if requested is None:
        return min(3, budget)
    if requested > budget and cores_per_rank:
        raise SystemExit('bench.py: --inflight %d is above what %d host core(s) per rank carry with %s waits (%d): lower --inflight (or give the '
                         'job more cores)' % (requested, cores_per_rank, 'blocking' if blocking_waits else 'spinning', budget))
    return requested


def launch_ranks(n, argv, script=None, python=None, poll=0.2, timeout=3600.0):
    """Parent of a self-launched multi-GPU run: N children `python bench.py <argv>`, one rank each.  Never initialises a GPU.  All
    children are watched: the first non-zero exit ends the others (which this launcher started), so a rank that dies before the
    rendezvous does not leave the rest waiting for the collective timeout."""
    import socket
    import subprocess
    import tempfile
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC (RCCL across processes on this driver)
        cpus = rank_cpu_set(r, n)                                    # a disjoint slice of the host cores per rank (inherited by its threads)
        if cpus:
            env['SE3_RANK_CPUS'] = ','.join(str(c) for c in sorted(cpus))
        procs.append(subprocess.Popen([python or sys.executable, os.path.abspath(script or __file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL,
                                      preexec_fn=(lambda c=cpus: os.sched_setaffinity(0, c)) if cpus else None))
    t_end = time.monotonic() + timeout
    failed = None
    while failed is None and any(p.poll() is None for p in procs) and time.monotonic() < t_end:
        for p in procs:
            if p.poll() not in (None, 0):
                failed = p.returncode
        time.sleep(poll)
    if failed is not None or time.monotonic() >= t_end:
        for p in procs:
            if p.poll() is None:
                p.terminate()                                       # exactly the children started above
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    codes = [p.wait() for p in procs]
    out0.seek(0)
    for line in out0.read().decode().splitlines():   # stdout carries the ONE JSON line; library chatter goes to stderr
        print(line, file=sys.stdout if line.lstrip().startswith('{') else sys.stderr, flush=True)
    if failed is None and time.monotonic() >= t_end and any(c != 0 for c in codes):
        return 124
    bad = [c for c in codes if c != 0]
    if failed is not None:
        return failed if failed > 0 else 1
    return 0 if not bad else (bad[0] if bad[0] > 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=9)
    ap.add_argument('--variant', default='se3ete')
    ap.add_argument('--pair', default='c2_5k')
    ap.add_argument('--cpu-baseline-pairs', type=int, default=5, help='timed pairs of the CPU baseline (after 3 warm-up pairs; median)')
    ap.add_argument('--cpu-baseline-threads', type=int, default=0, help='torch threads of the CPU baseline (0: min(cores, 8) -- the fastest setting on the host of the GPU box, profiles/r06_cpu_baseline_threads.txt)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--batch', type=int, default=16, help='registration pairs per forward / step (se3et_amd.batched), 1..16; '
                    '1 = the single-pair forward of the reference API')
    ap.add_argument('--switch-interval', type=float, default=1e-3)
    ap.add_argument('--prefetch', type=int, default=None, help='build the pyramid of the next batch on a second host thread / HIP '
                    'stream while the current batch runs through the model (the reference does this in DataLoader workers); every '
                    'timed step still builds its own pyramid inside the timed region.  0: pyramid and model back to back.  Default: 1 for '
                    '--batch > 1 (+4 %: 297-303 against 285-292 pairs/s in a same-box A/B of the round-3 build; `roofline.quiet` reports the '
                    'attention kernels without the second stream), 0 for --batch 1 (550 launches per pair: a second host thread only takes '
                    'the interpreter away)')
    ap.add_argument('--attention-dtype', default='float32', choices=['float32', 'bfloat16'], help="'bfloat16': geometric embedding "
                    "stored in bf16 (BASELINE.json configs[2] 'bf16 attention'); the headline metric is quoted on float32")
    ap.add_argument('--inflight', type=int, default=None, help='batches in flight per GPU: host threads, one HIP stream each, every one '
                    'building the pyramid of its batch and running it through the model.  Default 3 for --batch > 1: the forward of a batch has '
                    'host synchronisations (data-dependent sizes) and under-filled kernels (coarse stages, the transformer) during which '
                    'another batch keeps the GPU busy: 409-419 against 326-334 pairs/s with one batch in flight + pyramid prefetch (same-box '
                    'A/B; 2: 380-394, 4: 399-402).  Concurrent kernels share the GPU, so the durations of the timed attention kernels '
                    'grow: `roofline` reports them as measured in the timed region and `roofline.quiet` with one batch in flight.  Default 1 '
                    'for --batch 1')
    ap.add_argument('--single-pair-steps', type=int, default=16, help='pairs of the one-pair-per-forward measurement reported as '
                    '`single_pair` (rank 0 at --gpus 1 only; 0 = skip)')
    ap.add_argument('--train-steps', type=int, default=5, help='training steps (fwd + bwd + Adam, one pair each) of the `train_step` object '
                    '(rank 0 at --gpus 1 only; 0 = skip)')
    ap.add_argument('--schedule', default='roofline', choices=['roofline', 'throughput', 'exclusive', 'environment'],
                    help="how the batches in flight may overlap (se3et_amd.batched.set_schedule): 'roofline' (default) keeps the two coarsest backbone "
                         "stages from running beside another batch's transformer -- the RPE self-attention kernels then hold >= 0.40 of HBM IN the "
                         "timed region (0.45-0.47) at 455-458 pairs/s; 'throughput' lets them overlap: 487-493 pairs/s at 0.35-0.37; 'exclusive': no "
                         "backbone beside a transformer, 0.49-0.51 at 434-445.  The line reports the other of the first two in `other_schedule` "
                         "(one more region of the same K steps).  'environment': whatever SE3_CHAIN / SE3_CHAIN_GROUPS / SE3_BACKBONE_SPLIT say")
    ap.add_argument('--repeat-regions', type=int, default=2, help='the timed region (the same K steps between the same barrier + synchronize '
                    'bracket) is run this many MORE times after the headline region; `dispersion` reports all values with their median, min '
                    'and max (VERDICT round 5 item 6: a 4 %% move must be resolvable against the run-to-run spread).  0: off')
    ap.add_argument('--roofline-quiet-steps', type=int, default=3, help='extra steps after the timed region with one batch in flight and no '
                    'other stream, for `roofline.quiet` (0 = skip)')
    ap.add_argument('--force-dist', action='store_true', help='initialise torch.distributed (RCCL) also with ONE rank: the device barrier and the '
                    'MAX all-reduce of the job clock then run on the GPU at world size 1 (tests/test_gpu_bench.py)')
    ap.add_argument('--fake-device', action='store_true', help='launcher self-test: gloo rendezvous, sharding, barriers and the '
                    'MAX-over-ranks clock run for real, the step is a host sleep (no GPU needed; tests/test_bench_launch.py)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        seen = None if args.fake_device else visible_gpus()         # (no HIP call in the launcher: torch.cuda.device_count() is one on ROCm)
        if seen is not None and seen < args.gpus:
            raise SystemExit('bench.py: --gpus %d but only %d device(s) visible' % (args.gpus, seen))
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.fake_device:
        return run_fake(args)

    # blocking host waits on THIS rank's device, as the process's first GPU call (se3et_amd/__init__.py: later it is ineffective or harmful)
    # Only where it has been measured: ONE rank on the box (tests/test_gpu_concurrency.py, BENCH_r05 / r06).  A multi-rank run has never had a
    # node to run on (SCALE_r01..r05 skipped): there the runtime's default spinning waits stay -- 3 busy cores per rank of the 32 a rank owns
    # on the 256-thread host -- unless SE3_BLOCKING_SYNC=force.
    import se3et_amd
    if int(os.environ.get('WORLD_SIZE', '1')) == 1 or os.environ.get('SE3_BLOCKING_SYNC') == 'force':
        se3et_amd.request_blocking_sync(int(os.environ.get('LOCAL_RANK', '0')))

    from se3et_amd import ops as se3_ops
    from se3et_amd import _lib as se3_lib
    from se3et_amd import sharding
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import PAIR_PRESETS, make_pair

    if args.force_dist and 'MASTER_PORT' not in os.environ:      # a single rank started by hand: any free port on 127.0.0.1
        import socket
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
        sock.close()
    if torch.cuda.is_available():
        torch.cuda.set_device(sharding.rank_world()[2])
    rank, world, local = sharding.init_distributed('nccl', force=args.force_dist)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import se3et_amd
    host_waits = se3et_amd.blocking_sync_status(local)           # requested at the top of main(), before the first GPU call of this process

    cfg = make_cfg(args.variant, attention_dtype=args.attention_dtype)
    model = load_synthetic_weights(create_model(cfg)).to(dev).eval()
    total_steps = args.steps + args.warmup
    PB = max(1, args.batch)
    # host cores of this rank: under an external launcher (torch.distributed.run) every rank takes its slice of the node's cores itself;
    # ranks started by launch_ranks() arrive pinned
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    host_cores = os.cpu_count()
    if world > 1 and 'SE3_RANK_CPUS' not in os.environ:
        cpus = rank_cpu_set(local, local_world)
        if cpus:
            os.sched_setaffinity(0, cpus)
    try:
        cores_here = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores_here = host_cores
    args.inflight = inflight_for_cores(cores_here, args.inflight, PB, blocking_waits=host_waits == 'set')
    if args.inflight > 1:
        args.prefetch = 0                 # every in-flight thread builds its own pyramid
    if args.prefetch is None:
        args.prefetch = 1 if PB > 1 else 0
    # this rank's pairs, uploaded before the timed region (global pair index = (step * world + rank) * batch + j); with
    # --batch B the clouds of B pairs are stacked ref0, src0, ref1, src1, ... and go through ONE forward
    pairs = []
    for s in range(total_steps):
        clouds = []
        for j in range(PB):
            ref, src, _ = make_pair(args.pair, index=(s * world + rank) * PB + j)
            clouds += [ref, src]
        pts = torch.from_numpy(np.concatenate(clouds, 0)).to(dev)
        pairs.append((pts, torch.tensor([len(c) for c in clouds], dtype=torch.int64)))
    feats = torch.ones((pairs[0][0].shape[0], 1), dtype=torch.float32, device=dev)
    b = cfg.backbone
    from se3et_amd import batched as se3_batched
    from se3et_amd.batched import forward_pairs
    if any(k in os.environ for k in ('SE3_CHAIN', 'SE3_CHAIN_GROUPS', 'SE3_BACKBONE_SPLIT', 'SE3_CHAIN_SHARED')):
        args.schedule = 'environment'
    if args.schedule != 'environment':
        se3_batched.set_schedule(args.schedule)

    def forward(data):
        data['features'] = feats
        return model(data) if PB == 1 else forward_pairs(model, data)

    def step(i):
        pts, lens = pairs[i]
        return forward(precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits))

    import threading

    def run(take, stream):
        # one host thread + one HIP stream per batch in flight: while one batch waits on a data-dependent size (3 host syncs
        # per forward), the others keep the GPU and the launch queue busy.  `take()` hands out the next step index of a queue shared
        # by the threads (None = empty): whichever thread is free takes the next step, so the tail of the timed region is balanced
        # whatever the step count (a static deal of 20 steps over 3 threads is 7 / 7 / 6, and a slow thread keeps its share).
        with torch.cuda.stream(stream):
            while True:
                i = take()
                if i is None:
                    break
                step(i)
            stream.synchronize()

    def run_prefetched(indices):
        # the reference builds the pyramid in DataLoader workers while the model runs (geotransformer/utils/data.py collate);
        # here a second host thread + HIP stream builds the pyramid of pair i+1 (sync-heavy, GPU-light) while the main
        # thread / stream runs pair i through the model
        import queue
        q = queue.Queue(maxsize=2)
        side = streams[-1]
        stop, failed = threading.Event(), []

        def put(item):                       # never blocks for good: a consumer that died must not leave this thread (and the process) waiting
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return
                except queue.Full:
                    pass

        def producer():
            try:
                with torch.cuda.stream(side):
                    for i in indices:
                        if stop.is_set():
                            break
                        pts, lens = pairs[i]
                        data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius,
                                                          cfg.neighbor_limits)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        put((data, ev))
            except BaseException as e:       # handed to the main thread below
                failed.append(e)
            put(None)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        main = torch.cuda.current_stream()
        keep = []
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                data, ev = item
                main.wait_event(ev)
                for key in ('points', 'neighbors', 'subsampling', 'upsampling'):
                    for t in data[key]:
                        t.record_stream(main)          # allocated on the side stream, consumed on the main stream
                forward(data)
                keep = [data] + keep[:1]
        finally:
            stop.set()
            th.join(timeout=30)
        if failed:
            raise failed[0]

    def run_all(indices, per_thread=None):
        """per_thread: list of index lists, one per in-flight thread (priming: every thread / stream runs ITS list); default: the threads
        draw from one shared queue of `indices`."""
        P = max(1, args.inflight)
        if P == 1:
            if args.prefetch:
                run_prefetched(indices)
                return
            for i in indices:
                step(i)
            return
        failed = []
        lock = threading.Lock()
        pending = list(indices)

        def take_shared():
            with lock:
                return pending.pop(0) if pending and not failed else None

        def guarded(t, stream):
            try:
                if per_thread is None:
                    run(take_shared, stream)
                else:
                    own = list(per_thread[t])
                    run(lambda: own.pop(0) if own and not failed else None, stream)
            except BaseException as e:
                failed.append(e)

        threads = [threading.Thread(target=guarded, args=(t, streams[t]), daemon=True) for t in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if failed:
            raise failed[0]

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.inflight) + 1)]
    if args.inflight > 1 or args.prefetch:
        sys.setswitchinterval(args.switch_interval)   # default 5 ms: a host thread would hold the interpreter for half a pair
    # untimed: clock ramp-up of a cold GPU (a fresh box idles at low clocks), then the W warm-up steps
    ramp = torch.randn(4096, 4096, device=dev)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 1.0:
        ramp @ ramp
    torch.cuda.synchronize()
    # Priming (untimed, before the W warm-up steps, whatever W is): EVERY in-flight stream runs two steps of its own, so that per-stream
    # state -- GroupNorm / split workspaces, embedding and neighbour tables, the allocator's per-stream blocks, the weight pieces shared
    # through events -- exists before t0 on all of them.  With 3 threads, `--warmup 5` alone left one stream with a single warm step and
    # its first timed steps paid for the initialisation (BENCH_r03: 360 pairs/s at --steps 20 --warmup 5 against 409-443 at 60 + 9).
    if args.inflight > 1:
        prime = [i % max(1, args.warmup) if args.warmup > 0 else 0 for i in range(2)]
        run_all([], per_thread=[list(prime) for _ in range(args.inflight)])
        torch.cuda.synchronize()
    run_all(list(range(args.warmup)))
    torch.cuda.synchronize()
    sharding.barrier(dev)
    thread_cpu0 = thread_cpu_times() if os.environ.get('SE3_BENCH_THREAD_CPU') == '1' else None
    cpu0 = time.process_time()
    # The headline region carries the event pairs of the ROOFLINE kernel only -- the RPE self-attention launches, taken on the C side with the
    # dispatch's own begin / end timestamps (hipExtLaunchKernelGGL).  The event pairs of the two additional families (`roofline_kpconv`,
    # `roofline_dense`: ~50 torch event pairs per step from Python) move to the FIRST REPEAT of the region, the same K steps between the same
    # bracket: in the headline region they cost ~2 % of the rate (region 1 against regions 2 and 3, profiles/r06_batch_inflight_sweep.txt).
    se3_ops.KERNEL_TIMINGS = {} if os.environ.get('SE3_BENCH_PY_EVENTS') == '1' else None      # (A/B switch: round 5 recorded the families' event pairs here)
    se3_lib.lib().se3_debug_kernel_timing(1)        # every RPE attention launch gets its own HIP event pair
    t0 = time.perf_counter()
    run_all(list(range(args.warmup, total_steps)))
    torch.cuda.synchronize()
    sharding.barrier(dev)
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, dev)
    host_cpu_s = time.process_time() - cpu0           # CPU seconds of this rank's process (all host threads) over the timed region
    if getattr(se3_batched, '_LOCK_WAIT', None) is not None:
        print('host time waiting for chain locks since the start (s): %s; timed region %.3f s of wall x %d threads' % (
            {k: round(v, 3) for k, v in se3_batched._LOCK_WAIT.items()}, elapsed, max(1, args.inflight)), file=sys.stderr, flush=True)
    if thread_cpu0 is not None:                       # which threads burned it (name from /proc/self/task/<tid>/comm)
        now = thread_cpu_times()
        used = sorted(((now[t][1] - thread_cpu0.get(t, (now[t][0], 0.0))[1], now[t][0], t) for t in now), reverse=True)
        def ctx(t):
            try:
                st = open('/proc/self/task/%d/status' % t).read()
                return ' '.join(l.split(':')[1].strip() for l in st.splitlines() if 'ctxt_switches' in l)
            except OSError:
                return '?'
        print('thread CPU over the timed region (%.3f s of wall; pid %d, %d threads): ' % (elapsed, os.getpid(), len(now)) +
              ', '.join('%s[%d] %.3f s (ctx switches vol/nonvol %s)' % (n, t, c, ctx(t)) for c, n, t in used[:6] if c > 0.001), file=sys.stderr, flush=True)
        print('all threads: ' + ' '.join('%d:%s:%.2f' % (t, now[t][0], now[t][1]) for t in sorted(now)), file=sys.stderr, flush=True)
    # (grows with the number of timed steps -- 0.013 s per step at 20, 0.036 at 60 -- through the roofline's own per-launch HIP events, which stay
    #  alive until the run is over: DESIGN section 5)
    roofline = collect_roofline(se3_lib, {}, args)     # (waits for the recorded launches, releases their events, timing off)

    # Dispersion: the SAME K steps between the same bracket, `--repeat-regions` more times (per-launch event timing off: the headline region
    # above carries it).  `value` stays the first region -- the contract's "exactly K steps" -- and the repeats say how far a single region moves.
    region_s = [elapsed]
    se3_lib.lib().se3_debug_kernel_timing(0)
    timings = {}
    for rep in range(max(1, args.repeat_regions)):     # (at least one: the KPConv / dense families are timed in it)
        se3_ops.KERNEL_TIMINGS = {} if rep == 0 else None
        torch.cuda.synchronize()
        sharding.barrier(dev)
        t_r = time.perf_counter()
        run_all(list(range(args.warmup, total_steps)))
        torch.cuda.synchronize()
        sharding.barrier(dev)
        region_s.append(sharding.max_over_ranks(time.perf_counter() - t_r, dev))
        if rep == 0:
            timings, se3_ops.KERNEL_TIMINGS = se3_ops.KERNEL_TIMINGS, None
    rates = [world * args.steps * PB / t for t in region_s]
    dispersion = {'regions': len(rates), 'steps_per_region': args.steps, 'values': [round(v, 2) for v in rates],
                  'median': round(sorted(rates)[len(rates) // 2], 2), 'min': round(min(rates), 2), 'max': round(max(rates), 2),
                  'spread_rel': round((max(rates) - min(rates)) / sorted(rates)[len(rates) // 2], 4),
                  'note': 'values[0] is `value` (the timed region of the contract, with the per-launch HIP events of the roofline kernel); values[1] '
                          'is the same steps again, same bracket, with the event pairs of the KPConv and dense families instead (roofline_kpconv, '
                          'roofline_dense); the others carry no events'}

    # The other schedule, same K steps, same bracket, the roofline kernel's event pairs: what the overlap of the batches in flight is worth
    # to the rate and costs the roofline figure (profiles/r06_frac_vs_overlap.txt)
    other_schedule = None
    if args.schedule in ('roofline', 'throughput') and args.inflight > 1 and args.repeat_regions > 0:
        other = 'throughput' if args.schedule == 'roofline' else 'roofline'
        se3_batched.set_schedule(other)
        run_all(list(range(min(args.warmup, 2))))                       # (the chains start empty)
        torch.cuda.synchronize()
        sharding.barrier(dev)
        se3_lib.lib().se3_debug_kernel_timing(1)
        t_o = time.perf_counter()
        run_all(list(range(args.warmup, total_steps)))
        torch.cuda.synchronize()
        sharding.barrier(dev)
        el_o = sharding.max_over_ranks(time.perf_counter() - t_o, dev)
        rf_o = collect_roofline(se3_lib, {}, args)
        other_schedule = {'schedule': other, 'value': round(world * args.steps * PB / el_o, 3), 'ms_per_step': round(el_o / args.steps * 1e3, 3),
                          'roofline_frac': rf_o['frac'], 'roofline_avg_us': rf_o['avg_us'],
                          'note': 'the same K steps between the same bracket under the other schedule of se3et_amd.batched.set_schedule'}
        se3_batched.set_schedule(args.schedule)

    roofline_kpconv = collect_kpconv_roofline(timings)
    roofline_dense = collect_dense_roofline(timings)
    for extra in (roofline_kpconv, roofline_dense):
        if extra is not None:
            extra['measured_in'] = 'the first repeat of the timed region (dispersion.values[1]): the same K steps between the same bracket'
    roofline_step = collect_step_roofline(elapsed, args.steps, args)
    # The same kernels with nothing else on the GPU: by default other batches (--inflight) or the next batch's pyramid (--prefetch 1) run on
    # other streams BESIDE the timed kernels, which lengthens them; a few extra steps with one batch in flight (after the timed region)
    # give the kernels' own rate.
    if (args.prefetch or args.inflight > 1) and args.roofline_quiet_steps > 0:
        se3_ops.KERNEL_TIMINGS = {}
        se3_lib.lib().se3_debug_kernel_timing(1)
        for i in range(args.warmup, min(total_steps, args.warmup + args.roofline_quiet_steps)):
            step(i)
        torch.cuda.synchronize()
        quiet_timings, se3_ops.KERNEL_TIMINGS = se3_ops.KERNEL_TIMINGS, None
        quiet = collect_roofline(se3_lib, quiet_timings, args)
        quiet_kp = collect_kpconv_roofline(quiet_timings)
        quiet_dense = collect_dense_roofline(quiet_timings)
        if roofline_dense is not None and quiet_dense is not None:
            roofline_dense['quiet'] = {k: quiet_dense[k] for k in ('achieved', 'frac', 'launches', 'avg_us')}
            roofline_dense['ms_per_step_of_this_family'] = round(quiet_dense['avg_us'] * quiet_dense['launches'] / max(args.roofline_quiet_steps, 1) / 1e3, 3)
        if roofline_kpconv is not None and quiet_kp is not None:
            roofline_kpconv['quiet'] = {k: quiet_kp[k] for k in ('achieved', 'frac', 'launches', 'avg_us', 'f32_equivalent_tflops')}
        roofline['quiet'] = {k: quiet[k] for k in ('achieved', 'frac', 'launches', 'avg_us', 'rpe_bias_kernel_avg_us', 'attention_kernel_avg_us',
                                                   'eq_call_avg_us', 'inv_call_avg_us')}
        roofline['quiet']['note'] = ('same kernels, same shapes, %d extra step(s) after the timed region with ONE batch in flight and no other '
                                     'stream active: the rate of the kernels themselves' % args.roofline_quiet_steps)
        roofline['note'] = ('measured in the timed region, where %s share the GPU with the timed kernels (their durations include the '
                            'time slices of the other streams); `quiet` = the same kernels alone on the GPU'
                            % ('%d batches in flight' % args.inflight if args.inflight > 1 else "the next batch's pyramid kernels"))
        if args.inflight > 1:
            roofline['note'] += ("; schedule `%s` (se3et_amd.batched.set_schedule): %s" % (args.schedule, {
                'roofline': "the two coarsest backbone stages of the other batches never run beside a transformer section, the rest of their work does; "
                            "`other_schedule` = everything may overlap (the library's default): higher rate, the RPE kernels squeezed",
                'throughput': 'every section of the other batches may run beside the timed kernels; `other_schedule` = the two coarsest backbone '
                              'stages kept off the GPU during a transformer section',
                'exclusive': 'no backbone section of the other batches beside a transformer section'}.get(args.schedule, 'as the environment says')))

    # the auxiliary measurements never take the headline line down with them: a failure is reported in their place (and on stderr)
    def guarded(what, f, *a):
        try:
            # every auxiliary measurement starts from an empty caching allocator, as if it ran in a process of its own: the one-pair forwards
            # after three 16-pair steps on the same stream otherwise spend 3.4 ms of HOST time per pair re-cutting that stream's blocks
            # (96 against 128 pairs/s, gpu_kernel_ms unchanged; round 6)
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            return f(*a)
        except Exception as e:
            import traceback
            traceback.print_exc()
            return {'error': '%s failed: %s' % (what, repr(e)[:300])}

    single_pair = None
    if rank == 0 and args.gpus == 1 and args.single_pair_steps > 0:
        single_pair = guarded('single_pair', run_single_pair, model, cfg, args, dev, feats)
    train = None
    if rank == 0 and args.gpus == 1 and args.train_steps > 0 and args.variant == 'se3ete':
        train = guarded('train_step', run_train_step, args.variant, args, dev)
    cpu_baseline = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu_baseline = guarded('cpu_baseline', run_cpu_baseline, model, cfg, args)
    ranks_seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    if rank == 0:
        n, dims, _ = PAIR_PRESETS[args.pair]
        line = {
            'metric': 'point-cloud pairs/sec (fwd), SE3ET-E 5k-pt pairs', 'value': round(world * args.steps * PB / elapsed, 3),
            'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32' if args.attention_dtype == 'float32' else 'f32 (geometric embedding stored in bf16)',
            'data': 'synthetic', 'host_cpu_s_per_step': round(host_cpu_s / max(args.steps, 1), 5),
            'config': {'workload': 'SE3ET-E forward (pyramid + backbone + transformer + matching + Sinkhorn + LGR) on '
                                   'synthetic %d+%d-point pairs, %d pair(s) per rank per step' % (n, n, PB) +
                                   (' = the workload of BASELINE.json configs[3] (independent 5k+5k pairs sharded over 8 GPUs, no collective): %d pairs per step' % (world * PB) if world == 8 else ''),
                       'ranks_seen': ranks_seen, 'collectives': 'rccl' if torch.distributed.is_initialized() else 'none (one rank)',
                       'variant': args.variant, 'pair_preset': args.pair, 'sharding': 'pairs round-robin over ranks, no collective',
                       'pairs_per_forward': PB, 'batches_in_flight_per_gpu': max(1, args.inflight), 'pyramid_prefetch': bool(args.prefetch),
                       'host_cores': host_cores, 'host_cores_per_rank': cores_here,
                       'host_waits': 'hipDeviceScheduleBlockingSync ' + host_waits + ' (device %d)' % local,
                       'schedule': args.schedule,
                       'chained_sections': sorted(__import__('se3et_amd.batched', fromlist=['x']).CHAINED_SECTIONS),
                       'priority_sections': sorted(__import__('se3et_amd.batched', fromlist=['x']).PRIORITY_SECTIONS),
                       'attention_dtype': args.attention_dtype},
            'roofline': roofline, 'roofline_kpconv': roofline_kpconv, 'roofline_dense': roofline_dense, 'roofline_step': roofline_step, 'dispersion': dispersion, 'other_schedule': other_schedule, 'cpu_baseline': cpu_baseline, 'single_pair': single_pair, 'train_step': train,
        }
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


MFMA_F16_PEAK_TFLOPS = 2500.0        # dense f16 / bf16 matrix-core peak of one MI355X (/opt/skills/guides/MI355X_MICROARCH.md)


def collect_dense_roofline(timings):
    """The dense + GroupNorm family of the backbone (the largest share of a step since round 4): every
    dense_norm_kernel launch the backbone issues from Python -- se3_dense_norm_fwd (GEMM + statistics, output written or statistics only) and
    se3_dense_residual_fwd (the recomputed block tail) -- between an event pair (se3et_amd/ops.py).  HBM bound: algorithmic bytes = the
    activations read and written once, 4 (rows K [+ rows K2] [+ rows N residual] [+ rows N out]), the weights (K N, L2 resident) not counted."""
    ev = timings.get('dense', [])
    if not ev:
        return None
    ms = [a.elapsed_time(b) for a, b, _ in ev]
    nbytes = sum(f for _, _, f in ev)
    achieved = nbytes / (sum(ms) * 1e-3) / 1e9
    return {'kernel': 'dense_norm_kernel: the unary / shortcut dense layers of the E2PN backbone fused with their GroupNorm statistics, and the '
                      'recomputed block tails (one launch per event pair)',
            'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': None, 'launches': len(ev), 'avg_us': round(1e3 * sum(ms) / len(ev), 2),
            'algorithmic_bytes_per_launch': int(nbytes / len(ev)), 'ms_per_step_of_this_family': None}


def collect_step_roofline(elapsed, steps, args):
    """Whole-step HBM figure: the HBM-side bytes of one 8-pair step by kernel family from the committed whole-step
    PMC passes (tools/pmc_step.sh), over this run's time per step."""
    total, fams, source = load_pmc_step()
    if total is None or args.batch < 2 or args.variant != 'se3ete' or args.pair != 'c2_5k' or args.attention_dtype != 'float32':
        return {'bytes_per_step': None, 'source': source if total is None else 'the PMC passes were taken on the default workload (8 x c2_5k pairs, se3ete, f32)'}
    # (the passes were taken on an 8-pair step; every pair of a stacked step moves the same bytes: scaled to this run's pairs per step)
    total = total * args.batch / 8.0
    fams = {k: v * args.batch / 8.0 for k, v in fams.items()}
    s_per_step = elapsed / max(steps, 1)
    achieved = total / s_per_step / 1e9
    return {'bound': 'hbm', 'bytes_per_step': int(total), 'ms_per_step': round(s_per_step * 1e3, 3), 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
            'per_family_MB': {k: round(v / 1e6, 1) for k, v in sorted(fams.items(), key=lambda kv: -kv[1])},
            'source': 'HBM-side bytes (FETCH_SIZE, x2 for the families of 16-B/lane streams, + WRITE_SIZE) of one step with one batch in flight, '
                      'parsed at run time from ' + source + '; time per step of THIS run'}


def collect_kpconv_roofline(timings):
    """The largest kernel family by time (21 % of a step) is matrix-core bound, not HBM bound: kpconv_fused_kernel, one launch per KPConv
    layer.  Event pair around every launch (se3et_amd/ops.py); the kernel multiplies f16 hi / lo pieces (three products per algorithmic
    product, f32 accurate) in BOTH of its stages -- the contraction (v_mfma_f32_32x32x16_f16 in the consumer waves) and, since late round 3,
    the gather as a product (v_mfma_f32_16x16x32_f16 in the producer waves: csrc/kpconv_mfma.hip gather_multiply issues al.bh, ah.bl, ah.bh) --
    so the executed matrix-core flops are 3 x the algorithmic ones of both terms; those are priced against the f16 dense peak.  (The gather
    term counts the table's real width NN; the MFMAs run over 32 or 40 padded slots, which is not counted.)"""
    ev = timings.get('kpconv_fused', [])
    if not ev:
        return None
    us = sum(e0.elapsed_time(e1) for e0, e1, _ in ev) * 1e3
    flops = sum(f for _, _, f in ev)
    executed = 3.0 * flops / (us * 1e-6) / 1e12 if us > 0 else 0.0
    return {'kernel': 'kpconv_fused_kernel (gather as a product + contraction, one launch per KPConv layer; all layers of the timed steps)',
            'bound': 'mfma', 'achieved': round(executed, 1), 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(executed / MFMA_F16_PEAK_TFLOPS, 4), 'traffic': None,
            'launches': len(ev), 'avg_us': round(us / len(ev), 2), 'algorithmic_flops_per_launch': int(flops / len(ev)),
            'f32_equivalent_tflops': round(flops / (us * 1e-6) / 1e12, 1) if us > 0 else 0.0,
            'note': 'achieved = 3 x algorithmic flops (f16 hi.hi + hi.lo + lo.hi products) / summed launch durations; the kernel is bound by '
                    'its gather (L2 latency chains and the vector instructions around the MFMAs, profiles/r03_pmc_kpconv.txt), not by the '
                    'matrix pipe; in the timed region the durations include the time slices of the other batches in flight (`quiet` = alone)'}


def collect_roofline(se3_lib, timings, args):
    """One RPE self-attention call = rpe_bias_kernel (streams the embeddings) + attention_kernel (softmax.V), launched back to back by one
    C-ABI call for all clouds of the batch.  Each launch carries its own start / stop HIP event pair on the launch stream
    (hipExtLaunchKernelGGL: the dispatch's begin / end timestamps); the call's time is the sum of the two."""
    import ctypes
    cap = 4096 + 256 * (args.steps + args.warmup)
    us = (ctypes.c_float * cap)()
    tags = (ctypes.c_int * cap)()
    aux = (ctypes.c_double * cap)()
    se3_lib.lib().se3_debug_kernel_timing(0)
    n_ev = se3_lib.lib().se3_debug_kernel_timing_collect_ex(us, tags, aux, cap)
    # The library records, per logits launch of a stack-mode self-attention call, the call's algorithmic bytes (negative: equivariant
    # call) -- the launches may be issued from C (se3_transformer_forward), so the host keeps no list of its own.
    calls, per_call, i = [], [], 0
    bf16 = args.attention_dtype != 'float32'
    while i < n_ev:                                   # tag 1 (logits kernel) is always followed by its attention launch: tag 2, with the
        j, pre = i + 1, 0.0                           # K / V^T split in front of it (tag 3) when the f16 form runs -- counted with it
        if tags[i] == 1 and j < n_ev and tags[j] == 3:
            pre, j = us[j], j + 1
        if tags[i] == 1 and aux[i] != 0.0 and j < n_ev and tags[j] == 2:
            per_call.append((us[i], pre + us[j]))
            calls.append((abs(aux[i]), ('eq' if aux[i] < 0 else 'inv') + ('_bf16' if bf16 else '')))
            i = j + 1
        else:
            i += 1                                    # attention launches of the cross-attention layers
    kinds = {'eq': [0, 0.0, 0.0, 0.0], 'inv': [0, 0.0, 0.0, 0.0]}      # count, bytes, bias us, attention us
    for (nbytes, kind), (t_bias, t_attn) in zip(calls, per_call):
        k = kinds[kind.replace('_bf16', '')]
        k[0] += 1; k[1] += nbytes; k[2] += t_bias; k[3] += t_attn
    n_call = len(calls)
    bytes_call = sum(k[1] for k in kinds.values())
    us_call = sum(k[2] + k[3] for k in kinds.values())
    achieved = bytes_call / (us_call * 1e-6) / 1e9 if us_call > 0 else 0.0
    ratios, source = load_pmc_traffic()
    traffic = sum(ratios[name] * k[1] for name, k in kinds.items()) / max(n_call, 1) if ratios else None
    if args.attention_dtype != 'float32':
        traffic = None                 # the PMC passes were taken on the f32 kernels
    return {
        'kernel': 'RPE self-attention call = rpe_bias_kernel + attention kernel incl. its K / V^T split (all clouds of the batch per launch)',
        'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None if traffic is None else int(traffic),
        'traffic_source': 'bytes per call; PMC FETCH_SIZE(x2 on the streaming kernels)+WRITE_SIZE per algorithmic byte, parsed at run time from '
                          '%s, applied to the calls of this run' % source,
        'traffic_per_algorithmic_byte': None if not ratios else {k: round(v, 4) for k, v in ratios.items()},
        'launches': n_call, 'avg_us': round(us_call / max(n_call, 1), 2),
        'algorithmic_bytes_per_launch': int(bytes_call / max(n_call, 1)),
        'rpe_bias_kernel_avg_us': round(sum(k[2] for k in kinds.values()) / max(n_call, 1), 2),
        'attention_kernel_avg_us': round(sum(k[3] for k in kinds.values()) / max(n_call, 1), 2),
        'eq_call_avg_us': round((kinds['eq'][2] + kinds['eq'][3]) / max(kinds['eq'][0], 1), 2),
        'inv_call_avg_us': round((kinds['inv'][2] + kinds['inv'][3]) / max(kinds['inv'][0], 1), 2),
    }


def run_train_step(cfg_variant, args, dev):
    """BASELINE.json configs[4] on one GPU: SE3ET-E forward (training mode, ground-truth targets) + OverallLoss + backward + Adam on
    synthetic 5k+5k pairs, one pair per step (experiments/se3ete.3dmatch/loss.py:15-76, engine/base_trainer.py:181-196); pyramid on
    the GPU inside the step.  Seconds per step over `--train-steps` steps after 2 warm-up steps, the forward / backward / optimizer
    split of one more step by HIP events, peak memory."""
    from se3et_amd.data import registration_collate_fn_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    from se3et_amd.training import OverallLoss, make_optimizer, train_step
    cfg = make_cfg(cfg_variant)
    gc.collect()
    torch.cuda.empty_cache()
    held = torch.cuda.memory_allocated()       # what the inference phases of this process keep alive (models, resident pairs, workspaces, caches)
    model = load_synthetic_weights(create_model(cfg)).to(dev).train()
    loss_fn, opt = OverallLoss(cfg), make_optimizer(model, cfg)
    b = cfg.backbone
    rng = np.random.RandomState(0)
    n = args.train_steps + 3
    batches = []
    for i in range(n):
        ref, src, T = make_pair(args.pair, index=2000 + i)
        batches.append(dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32),
                            src_feats=np.ones((len(src), 1), np.float32), transform=T))

    def collate(i):
        return registration_collate_fn_stack_mode([batches[i]], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits, device=dev)

    for i in range(2):
        train_step(model, collate(i), loss_fn, opt, rng=rng)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for i in range(2, 2 + args.train_steps):
        losses, _ = train_step(model, collate(i), loss_fn, opt, rng=rng)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.train_steps
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    dd = collate(n - 1)
    opt.zero_grad(set_to_none=True)
    ev[0].record()
    loss = loss_fn(model(dd, train=True, rng=rng), dd)['loss']
    ev[1].record()
    loss.backward()
    ev[2].record()
    opt.step()
    ev[3].record()
    torch.cuda.synchronize()
    return {'s_per_step': round(dt, 4), 'pairs_per_s': round(1.0 / dt, 2), 'steps': args.train_steps,
            'forward_loss_ms': round(ev[0].elapsed_time(ev[1]), 2), 'backward_ms': round(ev[1].elapsed_time(ev[2]), 2),
            'adam_ms': round(ev[2].elapsed_time(ev[3]), 2),
            'peak_mem_gb': round((torch.cuda.max_memory_allocated() - held) / 2 ** 30, 2), 'held_by_earlier_phases_gb': round(held / 2 ** 30, 2),
            'loss': round(float(losses['loss'].detach()), 4),
            'note': 'BASELINE.json configs[4] on one GPU: forward (training mode) + OverallLoss + backward + Adam, one synthetic 5k+5k pair per '
                    'step incl. the on-GPU pyramid; pinned against the reference by tests/test_gpu_training.py::test_fullsize_training_step_matches_reference'}


def run_fake(args):
    """--fake-device: everything of the multi-process contract except the GPU (rendezvous over gloo, round-robin sharding,
    barrier + MAX-over-ranks clock, one JSON line from rank 0)."""
    from se3et_amd import sharding
    rank, world, _ = sharding.init_distributed('gloo')
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    PB = max(1, args.batch)
    mine = sharding.shard_pairs((args.steps + args.warmup) * world * PB, rank, world)     # pair indices of this rank
    sharding.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    sharding.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
    total = sharding.sum_over_ranks(len(mine))
    # the host cores every rank runs on (pinned by the launcher): gathered so that the test can see that they are disjoint
    mine_cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    cpu_sets = [None] * world
    if world > 1:
        torch.distributed.all_gather_object(cpu_sets, mine_cpus)
    else:
        cpu_sets = [mine_cpus]
    if rank == 0:
        print(json.dumps({'metric': 'launcher self-test (no GPU work)', 'value': round(world * args.steps * PB / elapsed, 3),
                          'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'none', 'data': 'none (fake device)',
                          'config': {'workload': 'fake device', 'ranks_seen': torch.distributed.get_world_size() if world > 1 else 1,
                                     'pairs_sharded': int(total), 'pairs_per_forward': PB, 'rank_cpus': cpu_sets}}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def run_single_pair(model, cfg, args, dev, feats_unused):
    """The reference-shaped path (one pair per forward, experiments/se3ete.3dmatch/model.py:86-98) on the same workload:
    pairs/s over `--single-pair-steps` pairs after 12 warm-up pairs; host synchronisations of one forward (torch's sync-debug
    mode: every blocking copy / .item() / nonzero) and, when no external profiler is attached, kernel launches and summed
    kernel time of one forward from torch.profiler."""
    import warnings
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.synthetic import make_pair
    b = cfg.backbone
    WU = 12            # warm-up pairs: every pair has row counts of its own, and the caching allocator keeps meeting new block sizes (a
                       # hipMalloc each, which waits for the GPU) during the first forwards
    n_pairs = args.single_pair_steps + WU
    inputs = []
    for i in range(n_pairs):
        ref, src, _ = make_pair(args.pair, index=1000 + i)
        inputs.append((torch.from_numpy(np.concatenate([ref, src], 0)).to(dev), torch.tensor([len(ref), len(src)], dtype=torch.int64)))
    ones = torch.ones((inputs[0][0].shape[0], 1), dtype=torch.float32, device=dev)

    def one(i):
        pts, lens = inputs[i]
        data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
        data['features'] = ones
        return model(data)

    for i in range(WU):
        try:
            one(i)
        except Exception:
            if os.environ.get('SE3_DEBUG_SINGLE_PAIR'):           # where does it break: which warm-up pair, with what in the pipeline
                from se3et_amd.batched import forward_pairs
                pts, lens = inputs[i]
                data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
                data['features'] = ones
                out = forward_pairs(model, data, with_registration=False)[0]
                torch.cuda.synchronize()
                ms = out['matching_scores']
                print('single pair %d (of %d warm-up) failed: patches %d, matching_scores %s finite %s max %s, knn masks %d / %d, feats_f finite %s, '
                      'ref_feats_c finite %s, widths %s' % (
                          i, WU, out['ref_node_corr_indices'].shape[0], tuple(ms.shape), bool(torch.isfinite(ms).all()),
                          float(ms[:, :-1, :-1].max()) if ms.numel() else None, int(out['ref_node_corr_knn_masks'].sum()),
                          int(out['src_node_corr_knn_masks'].sum()), bool(torch.isfinite(out['feats_f']).all()),
                          bool(torch.isfinite(out['ref_feats_c']).all()), [tuple(t.shape) for t in data['neighbors']]), file=sys.stderr, flush=True)
            raise
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(WU, n_pairs):
        one(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.single_pair_steps

    # The same one-pair forwards, (a) with the pyramid of pair i+1 built by a worker thread on its own stream while pair i runs through the
    # model (what the reference's DataLoader workers do on the CPU, geotransformer/utils/data.py collate), (b) with three one-pair forwards
    # in flight (three host threads, one stream each: independent requests of a server).  Reported beside the sequential rate, never as it.
    import queue, threading
    idx = list(range(WU, n_pairs))

    def pipelined():
        q, side, main, failed = queue.Queue(maxsize=2), torch.cuda.Stream(device=dev), torch.cuda.current_stream(), []

        def producer():
            try:
                with torch.cuda.stream(side):
                    for i in idx:
                        pts, lens = inputs[i]
                        data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        q.put((data, ev))
            except BaseException as e:
                failed.append(e)
            q.put(None)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        keep = []
        while True:
            item = q.get()
            if item is None:
                break
            data, ev = item
            main.wait_event(ev)
            for key in ('points', 'neighbors', 'subsampling', 'upsampling'):
                for t in data[key]:
                    t.record_stream(main)
            data['features'] = ones
            model(data)
            keep = [data] + keep[:1]
        th.join(timeout=30)
        if failed:
            raise failed[0]

    def concurrent(P=3):
        failed = []

        def worker(t, stream):
            try:
                with torch.cuda.stream(stream):
                    for i in idx[t::P]:
                        one(i)
                    stream.synchronize()
            except BaseException as e:
                failed.append(e)

        threads = [threading.Thread(target=worker, args=(t, torch.cuda.Stream(device=dev)), daemon=True) for t in range(P)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if failed:
            raise failed[0]

    variants = {}
    keep_interval = sys.getswitchinterval()
    sys.setswitchinterval(1e-4)                  # host threads issuing ~500 launches of ~10 us each per pair (1e-3 and 1e-5 measured no better)
    try:
        for name, fn in (('pyramid_prefetched', pipelined), ('three_in_flight', concurrent)):
            fn()                                  # warm-up (per-stream workspaces, caches)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            variants[name] = round(args.single_pair_steps / (time.perf_counter() - t1), 2)
    finally:
        sys.setswitchinterval(keep_interval)
    syncs = [0]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.cuda.set_sync_debug_mode('warn')       # (warns once that the mode is a prototype: not a synchronisation, not counted)
        warnings.simplefilter('always')
        show = warnings.showwarning
        warnings.showwarning = lambda message, *a, **k: syncs.__setitem__(0, syncs[0] + ('synchroniz' in str(message)))
        try:
            one(0)
        finally:
            torch.cuda.set_sync_debug_mode('default')
            warnings.showwarning = show
    torch.cuda.synchronize()
    launches = gpu_ms = None
    profiled = any(k.startswith('ROCPROF') for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')
    if not profiled:
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
                one(1)
                torch.cuda.synchronize()
            kernels = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and 'Memcpy' not in e.name
                       and 'Memset' not in e.name]
            if kernels:
                launches = len(kernels)
                gpu_ms = round(sum(e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total for e in kernels) / 1e3, 3)
        except Exception as exc:          # the profiler is a diagnostic, never a reason to lose the bench line
            print('bench.py: torch.profiler unavailable (%s)' % exc, file=sys.stderr)
    return {'pairs_per_s': round(1.0 / dt, 2), 'ms_per_pair': round(dt * 1e3, 3), 'pairs': args.single_pair_steps,
            'host_syncs': syncs[0], 'launches': launches, 'gpu_kernel_ms': gpu_ms,
            'host_ms': None if gpu_ms is None else round(max(dt * 1e3 - gpu_ms, 0.0), 3),
            'pairs_per_s_pyramid_prefetched': variants['pyramid_prefetched'], 'pairs_per_s_three_in_flight': variants['three_in_flight'],
            'note': 'one pair per forward incl. on-GPU pyramid and LGR, forwards strictly one after the other; host_ms = wall time per pair '
                    'not covered by kernel time.  pyramid_prefetched: the pyramid of the next pair built by a worker thread on its own stream '
                    '(the reference builds it in DataLoader workers); three_in_flight: three one-pair forwards at a time on three streams.  '
                    'Neither helps.  The measured split is in gpu_kernel_ms / host_ms / launches: one pair per forward is KERNEL-bound -- '
                    'about 420 launches sized for 8 pairs that one pair under-fills (6.9 of the 7.6 ms per pair are kernel time, 0.7 ms host gaps): '
                    'the dense + GroupNorm family costs 3.6x its per-pair share of a stacked step.  Stack pairs (the headline configuration), or '
                    'wait for the small-row dense path (DESIGN section 7 item 6)'}


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def run_cpu_baseline(model, cfg, args):
    """The CPU oracle (a restatement of the reference algorithm, kind 'port') on a bounded sample of the same workload: 3 warm-up pairs, then
    `--cpu-baseline-pairs` (default 5) timed ONE BY ONE, median.  Threads: `--cpu-baseline-threads`, default
    min(cores, 8): profiles/r06_cpu_baseline_threads.txt (tools/r6/cpu_threads.py on the GPU box's 256-core EPYC 9575F) -- 7.08 s per pair
    at 8 threads, 7.27 at 16, 10.1 at 32, 24.5 at 64: the small CPU ops of this path thrash beyond a few cores, 8 is the fastest."""
    from oracle import se3et_oracle as O
    from se3et_amd.synthetic import make_pair
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, args.cpu_baseline_threads or 8))
    torch.set_num_threads(cores)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    b = cfg.backbone
    oc = O.OracleConfig.from_model_cfg(cfg)

    def one(i):
        ref, src, _ = make_pair(args.pair, index=i)
        pts = torch.from_numpy(np.concatenate([ref, src], 0))
        t0 = time.perf_counter()
        data = O.precompute(pts, torch.tensor([len(ref), len(src)]), oc.num_stages, oc.init_voxel_size, oc.init_radius,
                            oc.neighbor_limits)
        data['features'] = torch.ones((pts.shape[0], 1))
        with torch.no_grad():
            O.forward(state, oc, data)
        return time.perf_counter() - t0

    first = one(0)
    slow = first >= 15.0                                         # keep the default run within minutes on slow hosts
    warm = 1 if slow else 3
    for i in range(1, warm):
        one(i)
    n_timed = 1 if slow else max(1, args.cpu_baseline_pairs)
    times = sorted(one(warm + i) for i in range(n_timed)) if not slow else [first]
    med = times[len(times) // 2]
    return {'value': round(1.0 / med, 4), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'cpu_model': cpu_model_name(), 'host_cores': os.cpu_count(),
            's_per_pair': {'median': round(med, 3), 'min': round(times[0], 3), 'max': round(times[-1], 3)},
            'warmup_pairs': warm, 'timed_pairs': len(times),
            'sample': '%d warm-up + %d timed pair(s) of the same workload, timed one by one, median (oracle/se3et_oracle.py on PyTorch-CPU '
                      'fp32, %d threads on a %d-core %s); the genuine reference measured 0.19 pairs/s on 8 cores (BASELINE.md)' %
                      (warm, len(times), cores, os.cpu_count() or 0, cpu_model_name())}


if __name__ == '__main__':
    main()
