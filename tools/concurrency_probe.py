"""Diagnostic: N rounds of `threads` batches in flight (one host thread + HIP stream each, as bench.py --inflight) against the same batches
run one after the other; prints, per round, which output tensors differ and by how much.  python tools/concurrency_probe.py [rounds] [threads] [pairs]"""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    thread_seq = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '3').split(',')]      # e.g. 2,3: rounds alternate 2 and 3 threads
    pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    from se3et_amd import ops
    from se3et_amd.batched import forward_pairs
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    cfg = make_cfg('se3ete')
    b = cfg.backbone
    for item in filter(None, os.environ.get('PROBE_FLAGS', '').split(',')):      # e.g. PROBE_FLAGS=GRAM_KERNEL=0,ATTENTION_F16=0 (se3et_amd.ops switches)
        name, val = item.split('=')
        from se3et_amd import functional as _SF
        mod = ops if hasattr(ops, name) else _SF
        setattr(mod, name, bool(int(val)))

    def batch(first):
        clouds = []
        for j in range(pairs):
            ref, src, _ = make_pair('c2_5k', index=first + j)
            clouds += [ref, src]
        pts = torch.from_numpy(np.concatenate(clouds, 0)).cuda()
        lens = torch.tensor([len(c) for c in clouds])

        def run():
            data = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
            data['features'] = torch.ones((pts.shape[0], 1), device='cuda')
            return data
        return run

    # optional trace of intermediate tensors (PROBE_TRACE=1): outputs of the geometric embedding, of every dense layer, of the attention
    # kernels' callers and of every add + LayerNorm, per host thread, in call order -- the first entry that differs names the racing op
    trace_on = os.environ.get('PROBE_TRACE', '0') == '1'
    tls = threading.local()
    if trace_on:
        from se3et_amd import functional as SF
        from se3et_amd import batched as BT

        def wrap(mod, name, tag=None):
            orig = getattr(mod, name)

            def f(*a, **k):
                out = orig(*a, **k)
                rec = getattr(tls, 'rec', None)
                if rec is not None:
                    outs = out if isinstance(out, (tuple, list)) else (out,)
                    for j, o in enumerate(outs):
                        if torch.is_tensor(o) and o.is_floating_point():
                            rec.append(('%s[%d]' % (tag or name, j), o.detach().clone()))
                return out
            setattr(mod, name, f)
        wrap(SF, 'geometric_embedding')
        wrap(SF, 'linear')
        wrap(SF, 'add_layer_norm')
        wrap(SF, 'anchor_max')
        wrap(ops, 'cross_attention_eq_stack')
        wrap(ops, 'rpe_self_attention_stack')
        wrap(ops, 'attention_stack')
        wrap(BT, '_cross_eq')
        wrap(BT, '_cross_plain')
    bad_rounds = 0
    always_fresh = os.environ.get('PROBE_FRESH', '0') == '1'
    for rnd in range(rounds):
        threads = thread_seq[rnd % len(thread_seq)]
        fresh = always_fresh or rnd % 2 == 0      # every other round on a fresh model with cold caches
        if fresh:
            model = load_synthetic_weights(create_model(cfg)).cuda().eval()
            ops.clear_weight_caches()
        builders = [batch(pairs * t + (0 if os.environ.get('PROBE_SAME', '0') == '1' else 10 * rnd)) for t in range(threads)]
        streams = [torch.cuda.Stream() for _ in range(threads)]
        got, failed = [None] * threads, []
        traces = [None] * threads
        gate = threading.Barrier(threads)

        def work(t):
            try:
                with torch.cuda.stream(streams[t]), torch.no_grad():
                    gate.wait()
                    tls.rec = [] if trace_on else None
                    outs = forward_pairs(model, builders[t]())
                    got[t] = [{k: v.detach().clone() for k, v in o.items() if torch.is_tensor(v)} for o in outs]
                    traces[t] = tls.rec
                    tls.rec = None
                    streams[t].synchronize()
            except BaseException as e:
                failed.append(e)

        pool = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for th in pool:
            th.start()
        for th in pool:
            th.join()
        if failed:
            raise failed[0]
        torch.cuda.synchronize()
        lines = []
        with torch.no_grad():
            for t in range(threads):
                tls.rec = [] if trace_on else None
                want = forward_pairs(model, builders[t]())
                if trace_on:
                    seq, tls.rec = tls.rec, None
                    shown = 0
                    for j, ((na, a), (nb, b_)) in enumerate(zip(traces[t], seq)):
                        if na != nb or a.shape != b_.shape or not torch.equal(a, b_):
                            d = (a.double() - b_.double()).abs() if a.shape == b_.shape else None
                            lines.append('  thread %d trace entry %d %s vs %s: %s' % (t, j, na, nb, 'shape' if d is None else
                                         'differs in %d of %d entries, max |d| %.3e (max |x| %.3e), first rows %s' % (
                                             int((d > 0).sum()), d.numel(), float(d.max()), float(b_.double().abs().max()),
                                             torch.nonzero(d.reshape(-1, d.shape[-1]).amax(1) > 0)[:4, 0].tolist())))
                            shown += 1
                            if shown >= 6:
                                break
                for p, (g, w) in enumerate(zip(got[t], want)):
                    for k in sorted(w):
                        if not torch.is_tensor(w[k]):
                            continue
                        if g[k].shape != w[k].shape:
                            lines.append('  thread %d pair %d %-28s shape %s vs %s' % (t, p, k, tuple(g[k].shape), tuple(w[k].shape)))
                        elif not torch.equal(g[k], w[k]):
                            d = (g[k].double() - w[k].double()).abs()
                            lines.append('  thread %d pair %d %-28s differs in %d of %d entries, max |d| %.3e (max |x| %.3e)'
                                         % (t, p, k, int((d > 0).sum()), d.numel(), float(d.max()), float(w[k].double().abs().max())))
        print('round %d (%d threads, %s model): %s' % (rnd, threads, 'fresh' if fresh else 'warm', 'identical' if not lines else '%d tensors differ' % len(lines)), flush=True)
        for ln in lines:
            print(ln, flush=True)
        bad_rounds += bool(lines)
    print('rounds with differences: %d of %d' % (bad_rounds, rounds))


if __name__ == '__main__':
    main()
