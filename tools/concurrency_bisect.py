"""Diagnostic for the intermittent first-use cross-stream difference ("Open issue"): every trial is a COLD start made
inside one process -- fresh model, weight caches cleared, per-stream workspaces / tables / plans dropped, the allocator's cache returned, new
streams -- a 2-thread round followed by a 3-thread round, compared with the same batches run one after the other.  Variants (round-robin over
the trials, so that a box's failure rate is sampled alike for all of them):

  base        nothing changed
  prewarm     one sequential forward on the main stream before the threads start (shared caches warm, streams cold)
  python      the transformer issued from Python (cdriver off)
  lock_tr     the transformer section of the threads serialised by a lock
  lock_bb     everything in front of the transformer serialised
  hostwait    readers of a shared cache entry wait for its event on the HOST
  warmws      the embedding's record workspaces survive the cold start

On a differing trial it prints which of {threaded, sequential, sequential again} is the outlier and the first stage whose checksum differs
(backbone output, each cloud's embedding, transformer output).   python tools/concurrency_bisect.py [trials per variant] [variants,comma]"""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    variants = (sys.argv[2] if len(sys.argv) > 2 else 'base,prewarm,python,lock_tr,lock_bb,hostwait').split(',')
    from se3et_amd import batched as BT
    from se3et_amd import cdriver, ops
    from se3et_amd import functional as SF
    from se3et_amd.data import precompute_data_stack_mode
    from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
    from se3et_amd.synthetic import make_pair
    cfg = make_cfg('se3ete')
    b = cfg.backbone
    tls = threading.local()
    KEEP = os.environ.get('BISECT_KEEP', '0') == '1'       # keep every embedding of both runs and describe where they differ
    tr_lock, bb_lock = threading.Lock(), threading.Lock()
    state = {'variant': 'base'}

    def chk(t):
        return t.detach().double().sum()

    # checksums of the stages, per thread, in call order (a reduction launch each: small next to the forward)
    orig_tp, orig_ge = BT.transformer_pairs, SF.geometric_embedding

    def transformer_pairs(gt, points_c, lengths_c, feats_c, packed=False):
        rec = getattr(tls, 'rec', None)
        if rec is not None:
            rec.append(('backbone', chk(feats_c)))
        held = None
        if state['variant'] == 'lock_bb' and getattr(tls, 'bb_held', False):
            bb_lock.release()
            tls.bb_held = False
        if state['variant'] == 'lock_tr' and getattr(tls, 'threaded', False):
            tr_lock.acquire()
            held = tr_lock
        try:
            out = orig_tp(gt, points_c, lengths_c, feats_c, packed=packed)
            if held is not None:
                torch.cuda.current_stream().synchronize()
        finally:
            if held is not None:
                held.release()
        if rec is not None:
            rec.append(('transformer', chk(out[0]) if packed else sum(chk(o) for o in out[0] + out[1])))
        return out

    def geometric_embedding(*a, **k):
        out = orig_ge(*a, **k)
        rec = getattr(tls, 'rec', None)
        if rec is not None:
            outs = out if isinstance(out, tuple) else (out,)
            rec.append(('embedding', sum(chk(o.float()) for o in outs if o is not None)))
            if KEEP:
                N_ = a[0].shape[0]
                wsb = ops._emb_ws[(a[0].device, ops._stream().value)][:N_ * N_ * 80].clone()
                tls.embs.append(tuple(None if o is None else o.detach().clone() for o in outs) + (k['knn'].clone(), a[0].clone(), wsb))
        return out
    BT.transformer_pairs, SF.geometric_embedding = transformer_pairs, geometric_embedding
    orig_get = ops._Shared.get

    def get_hostwait(self):
        self.event.synchronize()
        return orig_get(self)

    def batch(first, pairs=2):
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

    def cold():
        ops.clear_weight_caches()
        for d in (ops._gn_workspace, ops._dense_ws, ops._kpconv_split_ws) + (() if state['variant'] == 'warmws' else (ops._emb_ws,)) + (ops._emb_table_cache, ops._neighbor_table_cache,
                  ops._pair_rows_cache, ops._nonzero_norm_cache, ops._host_table_cache, BT._pack_index_cache, cdriver._ws):
            d.clear()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    builders_all = [batch(2 * t) for t in range(3)]
    counts = {v: [0, 0] for v in variants}
    for trial in range(trials * len(variants)):
        variant = variants[trial % len(variants)]
        state['variant'] = variant
        cdriver.ENABLED = variant != 'python'
        ops._Shared.get = get_hostwait if variant == 'hostwait' else orig_get
        cold()
        model = load_synthetic_weights(create_model(cfg)).cuda().eval()
        if variant == 'prewarm':
            with torch.no_grad():
                BT.forward_pairs(model, builders_all[0]())
            torch.cuda.synchronize()
        report = []
        for threads in (2, 3):
            builders = builders_all[:threads]
            streams = [torch.cuda.Stream() for _ in range(threads)]
            got, recs, failed, kept = [None] * threads, [None] * threads, [], [None] * threads
            gate = threading.Barrier(threads)

            def work(t):
                try:
                    with torch.cuda.stream(streams[t]), torch.no_grad():
                        gate.wait()
                        tls.rec, tls.threaded, tls.embs = [], True, []
                        if variant == 'lock_bb':
                            bb_lock.acquire()
                            tls.bb_held = True
                        outs = BT.forward_pairs(model, builders[t]())
                        got[t] = [{k: v.detach().clone() for k, v in o.items() if torch.is_tensor(v)} for o in outs]
                        recs[t], tls.rec, tls.threaded = tls.rec, None, False
                        kept[t] = tls.embs
                        streams[t].synchronize()
                except BaseException as e:
                    failed.append(e)
                finally:
                    if getattr(tls, 'bb_held', False):
                        bb_lock.release()
                        tls.bb_held = False

            pool = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
            for th in pool:
                th.start()
            for th in pool:
                th.join()
            if failed:
                raise failed[0]
            torch.cuda.synchronize()
            all_seq, all_want, all_rec = {}, {}, {}
            with torch.no_grad():
                for t in range(threads):
                    tls.rec, tls.embs = [], []
                    all_want[t] = BT.forward_pairs(model, builders[t]())
                    all_rec[t], tls.rec = tls.rec, None
                    all_seq[t] = tls.embs
                for t in range(threads):
                    want, seq, seq_embs = all_want[t], all_rec[t], all_seq[t]
                    same = all(torch.equal(g[k], w[k]) for g, w in zip(got[t], want) for k in ('ref_feats_c', 'src_feats_c')
                               if g[k].shape == w[k].shape) and all(g['ref_feats_c'].shape == w['ref_feats_c'].shape for g, w in zip(got[t], want))
                    if same:
                        continue
                    tls.rec, tls.embs = [], []
                    again = BT.forward_pairs(model, builders[t]())
                    seq2, tls.rec = tls.rec, None
                    seq_same = all(torch.equal(a[k], w[k]) for a, w in zip(again, want) for k in ('ref_feats_c', 'src_feats_c'))
                    first = next(('%d:%s' % (j, na) for j, ((na, a), (_, c)) in enumerate(zip(recs[t], seq)) if float(a) != float(c)), 'none')
                    pairs_bad = [p for p, (g, w) in enumerate(zip(got[t], want)) if not torch.equal(g['src_feats_c'], w['src_feats_c'])
                                 or not torch.equal(g['ref_feats_c'], w['ref_feats_c'])]
                    dmax = max(float((g[k] - w[k]).abs().max()) for g, w in zip(got[t], want) for k in ('ref_feats_c', 'src_feats_c'))
                    if KEEP:
                        for c, (ga, sa) in enumerate(zip(kept[t], seq_embs)):
                            if not torch.equal(ga[4], sa[4]):
                                N_ = ga[3].shape[0]
                                gj, sj = ga[4][:N_ * N_ * 16].view(torch.int32).view(-1, 4), sa[4][:N_ * N_ * 16].view(torch.int32).view(-1, 4)
                                gw, sw = ga[4][N_ * N_ * 16:].view(torch.float32).view(-1, 4, 4), sa[4][N_ * N_ * 16:].view(torch.float32).view(-1, 4, 4)
                                badj = torch.nonzero((gj != sj).any(1))[:, 0]
                                badw = torch.nonzero((gw != sw).any(2).any(1))[:, 0]
                                report.append('  cloud %d records: %d pairs with other intervals %s, %d pairs with other weights %s' % (
                                    c, len(badj), badj[:6].tolist(), len(badw), badw[:6].tolist()))
                                for r in (badw[:3].tolist() + badj[:2].tolist()):
                                    report.append('    pair %d (n %d m %d): intervals got %s want %s; weights got %s want %s' % (
                                        r, r // N_, r % N_, gj[r].tolist(), sj[r].tolist(), [[round(v, 4) for v in row] for row in gw[r].tolist()],
                                        [[round(v, 4) for v in row] for row in sw[r].tolist()]))
                            for nm, g_, s_ in zip(('emb', 'eq', 'knn', 'pts'), ga, sa):
                                if g_ is None or torch.equal(g_, s_):
                                    continue
                                d = (g_.double() - s_.double()).abs()
                                if nm == 'emb':
                                    rows = torch.nonzero(d.amax((1, 2)) > 0)[:, 0]
                                    cols = torch.nonzero(d.amax((0, 2)) > 0)[:, 0]
                                    chans = torch.nonzero(d.amax((0, 1)) > 0)[:, 0]
                                    # whose records did the bad blocks use?  compare each bad 16-pair block with the same flat pair index of every
                                    # other embedding (this thread's other clouds share the record workspace; other threads have their own)
                                    C_ = g_.shape[-1]
                                    gf, sf = g_.reshape(-1, C_), s_.reshape(-1, C_)
                                    bad = torch.nonzero((gf != sf).any(1))[:, 0]
                                    blocks = sorted(set((bad // 16).tolist()))
                                    found = {}
                                    for bk in blocks[:64]:
                                        blkv = gf[16 * bk:16 * bk + 16]
                                        hit = 'unknown'
                                        if float(blkv.abs().max()) == 0.0:
                                            hit = 'zeros'
                                        for t2 in range(threads):
                                            for c2, other in enumerate(all_seq[t2] if t2 in all_seq else []):
                                                of = other[0].reshape(-1, C_)
                                                if (t2, c2) != (t, c) and of.shape[0] >= 16 * bk + 16 and torch.equal(of[16 * bk:16 * bk + 16][:blkv.shape[0]], blkv):
                                                    hit = 'thread %d cloud %d' % (t2, c2)
                                        found[hit] = found.get(hit, 0) + 1
                                    report.append('  bad blocks of thread %d cloud %d (of %d): same flat index in %s' % (t, c, len(blocks), found))
                                    bk = blocks[0]
                                    for r in range(16 * bk, min(16 * bk + 16, gf.shape[0]), 5):
                                        same_row = torch.nonzero((sf == gf[r]).all(1))[:, 0].tolist()[:4]
                                        elsewhere = []
                                        for t2 in range(threads):
                                            for c2, other in enumerate(all_seq[t2]):
                                                if (t2, c2) != (t, c):
                                                    idx = torch.nonzero((other[0].reshape(-1, C_) == gf[r]).all(1))[:, 0].tolist()[:2]
                                                    if idx:
                                                        elsewhere.append((t2, c2, idx))
                                        report.append('    pair %d (n %d m %d): got %s want %s; rows of this cloud with the got values: %s; of other clouds: %s; '
                                                      'channels equal %d of %d' % (r, r // g_.shape[1], r % g_.shape[1], [round(v, 3) for v in gf[r, :6].tolist()],
                                                                                  [round(v, 3) for v in sf[r, :6].tolist()], same_row, elsewhere,
                                                                                  int((gf[r] == sf[r]).sum()), C_))
                                    report.append('  cloud %d emb %s: %d entries differ, max %.2e; rows n %d [%d..%d], cols m %d [%d..%d], channels %d %s'
                                                  % (c, tuple(g_.shape), int((d > 0).sum()), float(d.max()), len(rows), int(rows.min()), int(rows.max()),
                                                     len(cols), int(cols.min()), int(cols.max()), len(chans), chans[:40].tolist()))
                                else:
                                    report.append('  cloud %d %s: %d entries differ, max %.2e' % (c, nm, int((d > 0).sum()), float(d.max())))
                    report.append('%d threads: thread %d pairs %s max|d| %.2e; sequential runs agree: %s; first differing stage %s; stages %s'
                                  % (threads, t, pairs_bad, dmax, seq_same, first, [n for n, _ in recs[t]][:8]))
        counts[variant][0] += 1
        counts[variant][1] += bool(report)
        print('trial %3d %-8s %s' % (trial, variant, 'identical' if not report else 'DIFFERS'), flush=True)
        for ln in report:
            print('    ' + ln, flush=True)
        del model
    for v in variants:
        print('%-8s: %d of %d trials differ' % (v, counts[v][1], counts[v][0]))


if __name__ == '__main__':
    main()
