"""Fixture generator (runs ONLY in the build container, where /root/reference exists).

Imports the genuine reference through oracle/ref_shims.py, runs it on deterministic synthetic inputs and
stores inputs + expected outputs as small .npz files next to this script.  Nothing of the reference's source
travels; the fixtures are data.  Re-run with:  python tests/golden/generate_golden.py

Fixtures
  tables_kanchor6.npz      constant anchor / permutation tables of the octahedral (kanchor=6, C4 quotient) setup
  precompute_c1.npz        stage pyramid (points, lengths, 10 neighbour index arrays) of the C1 2k+2k pair
  precompute_sizes.npz     stage lengths / neighbour widths for the C2 (5k) and C3 (20k, KITTI cfg) pairs
  micro_se3ete.npz / micro_se3eti.npz
                           micro-config full model: pair, reference-initialised state dict (seed 0), collated
                           pyramid, per-op inputs/outputs captured with forward hooks, and model outputs
  synthw_<variant>.npz     real-width configs with name-keyed synthetic weights (se3et_amd.synthetic.synth_tensor):
                           state-dict names/shapes + outputs only (weights are regenerated from the names)
  train_micro_se3ete.npz / train_micro_se3eti.npz
                           training step of the micro models (configs[4] pieces): losses, ground-truth correspondences, gradient
                           norms of all parameters, a few complete gradients, parameter checksum after one Adam step
  vgtk_ops.npz             EPN toolkit ops through their importable PyTorch twins (inter grouping fwd + grad, gather)
  c2_se3ete_5k.npz         BASELINE.json configs[1] at FULL size: SE3ET-E on the 5k+5k pairs 0..7 (the pairs of one bench step),
                           synthetic weights; per-layer outputs and index-table checksums for pair 0
  c3_se3eti_kitti_20k.npz  BASELINE.json configs[2] at FULL size: SE3ET-I (KITTI configuration) on the 20k+20k pair
  demo_se3ete.npz          the reference's real pair data/demo/{ref,src,gt}.npy (demo.py:44-58) through the genuine collate and SE3ET-E
  table_tiecanon.npz       tie-canonical checksums of the reference's tables for the clouds that hold exact distance ties (C3, cap pair)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
from helpers import tie_canonical  # noqa: E402  (tests/helpers.py: shared with the tests that compare against these checksums)
from se3et_amd.synthetic import make_pair, synth_tensor  # noqa: E402

LEARNED_LEAVES = ('weight', 'bias', 'weights', 'alpha')


def _np(t):
    t = t.detach().cpu()
    if t.dtype == torch.int64:
        return t.numpy().astype(np.int32) if t.numel() and int(t.abs().max()) < 2 ** 31 else t.numpy()
    return t.numpy()


def collate(ref, src, T, num_stages, voxel, radius, limits):
    from geotransformer.utils.data import registration_collate_fn_stack_mode
    d = dict(ref_points=ref, src_points=src, ref_feats=np.ones((len(ref), 1), np.float32),
             src_feats=np.ones((len(src), 1), np.float32), transform=T)
    return registration_collate_fn_stack_mode([d], num_stages, voxel, radius, limits)


def pack_pyramid(dd, out, prefix='data/'):
    for key in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
        for i, t in enumerate(dd[key]):
            out['%s%s_%d' % (prefix, key, i)] = _np(t)


def gen_tables():
    ref_shims.install()
    import geotransformer.modules.transformer.utils_epn.anchors as L
    import vgtk.functional as fr
    from geotransformer.modules.e2pn.blocks_epn import KPConvInterSO3
    vs, v_adjs, vRs, ecs, face_normals = L.get_octahedron_vertices()
    trace_ori, trace_rot = fr.get_relativeV_index(vRs, vs)
    conv = KPConvInterSO3(15, 6, 4, 4, 1.0, 1.0, equiv_mode_kp=True, non_sep_conv=True, rot_by_permute=True,
                          fixed_kernel_points='center', quotient_factor=4)
    import e3nn.o3 as o3
    anchors = torch.tensor(L.get_anchorsV24(), dtype=torch.float32)
    np.savez_compressed(
        os.path.join(HERE, 'tables_kanchor6.npz'),
        vertices=vs, vRs=vRs, anchors=L.get_anchorsV24(), face_normals=face_normals, v_adj0=np.int64(v_adjs[0, 0]),
        trace_idx_ori=trace_ori, trace_idx_rot=trace_rot, quotient_anchors=_np(conv.quotient_anchors),
        kernel_points_unit=_np(conv.kernel_points), kidx_rot=_np(conv.kidx_rot), ridx_rot=_np(conv.ridx_rot),
        conv_anchors=_np(conv.anchors),
        wignerD0=_np(o3.Irrep(0, 1).D_from_matrix(anchors.transpose(1, 2))),
        wignerD1=_np(o3.Irrep(1, 1).D_from_matrix(anchors.transpose(1, 2))))


def gen_precompute():
    ref_shims.install()
    ref, src, T = make_pair('c1_2k')
    dd = collate(ref, src, T, 4, 0.025, 0.0625, [38, 36, 36, 38])
    out = {'ref': ref, 'src': src}
    pack_pyramid(dd, out, '')
    np.savez_compressed(os.path.join(HERE, 'precompute_c1.npz'), **out)
    sizes = {}
    for name, stages, voxel, radius, limits in (('c2_5k', 4, 0.025, 0.0625, [38, 36, 36, 38]),
                                                ('c3_20k', 5, 0.3, 1.275, [38, 36, 36, 38, 38])):
        ref, src, T = make_pair(name)
        dd = collate(ref, src, T, stages, voxel, radius, limits)
        sizes[name + '/lengths'] = np.stack([_np(l) for l in dd['lengths']])
        sizes[name + '/neighbor_widths'] = np.array([n.shape[1] for n in dd['neighbors']])
        sizes[name + '/subsampling_widths'] = np.array([n.shape[1] for n in dd['subsampling']])
        sizes[name + '/upsampling_widths'] = np.array([n.shape[1] for n in dd['upsampling']])
        # order-sensitive checksums of the index arrays (sum of value * (position + 1) mod 2^61-1)
        for key in ('neighbors', 'subsampling', 'upsampling'):
            cs = []
            for t in dd[key]:
                v = t.numpy().astype(np.uint64).reshape(-1)
                w = (np.arange(v.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
                cs.append(int((v * w).sum() % np.uint64(2 ** 61 - 1)))
            sizes[name + '/' + key + '_checksum'] = np.array(cs, dtype=np.uint64)
        sizes[name + '/points_last'] = _np(dd['points'][-1])
    np.savez_compressed(os.path.join(HERE, 'precompute_sizes.npz'), **sizes)


def _hook(store, name, keep_inputs=True):
    def f(mod, args, kwargs, output):
        if name in store:          # the first call only (ref <- ref / ref <- src direction)
            return
        rec = {}
        if keep_inputs:
            for i, a in enumerate(args):
                if torch.is_tensor(a):
                    rec['in%d' % i] = _np(a)
            for k, a in kwargs.items():
                if torch.is_tensor(a):
                    rec['kw_' + k] = _np(a)
        outs = output if isinstance(output, (tuple, list)) else (output,)
        flat = []
        for o in outs:
            if isinstance(o, (tuple, list)):
                flat.extend(o)
            else:
                flat.append(o)
        for i, o in enumerate(flat):
            if torch.is_tensor(o):
                rec['out%d' % i] = _np(o)
        store[name] = rec
    return f


def run_model(variant, micro, synth_seed=None, pair='micro', index=0, light=False):
    make_cfg, create_model = ref_shims.load_experiment(variant)
    cfg = make_cfg()
    if micro:
        cfg.backbone.init_dim, cfg.backbone.output_dim, cfg.backbone.group_norm = 8, 32, 4
        cfg.geotransformer.input_dim, cfg.geotransformer.hidden_dim, cfg.geotransformer.output_dim = 128, 32, 32
    torch.manual_seed(0)
    model = create_model(cfg).eval()
    if synth_seed is not None:
        sd = model.state_dict()
        for k in sd:
            if k.rsplit('.', 1)[-1] in LEARNED_LEAVES and sd[k].dtype == torch.float32 and 'anchors' not in k:
                sd[k] = torch.from_numpy(np.asarray(synth_tensor(k, sd[k].shape, synth_seed)))
        model.load_state_dict(sd)
    ref, src, T = pair if isinstance(pair, tuple) else make_pair(pair, index=index)
    dd = collate(ref, src, T, cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius,
                 [38, 36, 36, 38, 38][:cfg.backbone.num_stages])
    ops = {}
    tr = model.transformer.transformer
    if light:          # full-size pairs: layer outputs only (the attention hooks would keep the (N, N, C) embeddings per layer)
        for i, layer in enumerate(tr.layers):
            layer.register_forward_hook(_hook(ops, 'layer_%d' % i, keep_inputs=False), with_kwargs=True)
        feats = {}
        model.backbone.register_forward_hook(lambda m, a, o: feats.update(c=o[-1], f=o[0]))
        with torch.no_grad():
            out = model(dd)
        return cfg, model, (ref, src, T), dd, ops, feats, out
    model.backbone.encoder2_2.interso3.conv.register_forward_hook(_hook(ops, 'kpconv_2_2'), with_kwargs=True)
    model.backbone.encoder2_1.interso3.conv.register_forward_hook(_hook(ops, 'kpconv_2_1'), with_kwargs=True)
    model.backbone.encoder1_1.register_forward_hook(_hook(ops, 'simple_1_1'), with_kwargs=True)
    model.backbone.encoder2_1.register_forward_hook(_hook(ops, 'resnet_2_1'), with_kwargs=True)
    model.transformer.embedding.register_forward_hook(_hook(ops, 'embedding'), with_kwargs=True)
    model.optimal_transport.register_forward_hook(_hook(ops, 'sinkhorn'), with_kwargs=True)
    model.coarse_matching.register_forward_hook(_hook(ops, 'coarse_matching'), with_kwargs=True)
    for i, layer in enumerate(tr.layers):
        layer.attention.attention.register_forward_hook(_hook(ops, 'attn_%d' % i), with_kwargs=True)
        layer.register_forward_hook(_hook(ops, 'layer_%d' % i, keep_inputs=False), with_kwargs=True)
    feats = {}
    model.backbone.register_forward_hook(lambda m, a, o: feats.update(c=o[-1], f=o[0]))
    with torch.no_grad():
        out = model(dd)
    return cfg, model, (ref, src, T), dd, ops, feats, out


def pack_outputs(out, feats, res, full):
    res['out/feats_c'] = _np(feats['c']) if full else _np(feats['c'])[:, :, :64]
    res['out/feats_f'] = _np(feats['f']) if full else _np(feats['f'])[::8]
    for k in ('ref_feats_c', 'src_feats_c', 'ref_node_corr_indices', 'src_node_corr_indices', 'estimated_transform',
              'corr_scores'):
        res['out/' + k] = _np(out[k])
    ms = out['matching_scores']
    res['out/matching_scores_head'] = _np(ms[:8])
    res['out/matching_scores_rowsum'] = _np(ms[:, :-1, :-1].exp().sum((1, 2)))
    res['out/num_corr'] = np.int64(out['ref_corr_points'].shape[0])


def gen_micro(variant, fname):
    cfg, model, (ref, src, T), dd, ops, feats, out = run_model(variant, micro=True)
    res = {'ref': ref, 'src': src, 'transform': T, 'blocks': np.array(cfg.geotransformer.blocks)}
    for k, v in model.state_dict().items():
        res['sd/' + k] = _np(v) if v.dtype != torch.int64 else v.numpy()
    pack_pyramid(dd, res)
    for name, rec in ops.items():
        for k, v in rec.items():
            if name.startswith('layer_') and k != 'out0':
                continue                      # layer hooks: output states only
            if name.startswith('attn_') and k in ('in3', 'kw_embed_eq'):
                continue                      # the embeddings are stored once (op/embedding/*)
            if name.startswith('attn_') and k == 'out1' and name not in ('attn_0', 'attn_4'):
                continue                      # score tensors: one equivariant + one invariant self layer only
            if name == 'sinkhorn':
                v = v[:16]
            res['op/%s/%s' % (name, k)] = v
    pack_outputs(out, feats, res, full=True)
    np.savez_compressed(os.path.join(HERE, fname), **res)


def gen_synthw(variant, fname, pair='micro', keep_ops=True):
    cfg, model, (ref, src, T), dd, ops, feats, out = run_model(variant, micro=False, synth_seed=7, pair=pair)
    res = {'blocks': np.array(cfg.geotransformer.blocks), 'pair': np.array(pair), 'synth_seed': np.int64(7)}
    names, shapes, dtypes = [], [], []
    for k, v in model.state_dict().items():
        names.append(k)
        shapes.append(','.join(str(s) for s in v.shape))
        dtypes.append(str(v.dtype).replace('torch.', ''))
    res['sd_names'], res['sd_shapes'], res['sd_dtypes'] = np.array(names), np.array(shapes), np.array(dtypes)
    res['lengths'] = np.stack([_np(l) for l in dd['lengths']])
    for i in (0, 3) if keep_ops else ():
        rec = ops.get('attn_%d' % i, {})
        if 'out0' in rec:
            res['op/attn_%d/out0' % i] = rec['out0']
    for name in ops if keep_ops else ():
        if name.startswith('layer_'):
            res['op/%s/out0' % name] = ops[name]['out0']
    pack_outputs(out, feats, res, full=False)
    np.savez_compressed(os.path.join(HERE, fname), **res)


def _index_checksum(t):
    """Order-sensitive checksum of an index array: sum of value * ((position mod 65521) + 1) mod 2^61 - 1 (as gen_precompute)."""
    v = np.asarray(t).astype(np.uint64).reshape(-1)
    w = (np.arange(v.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return int((v * w).sum() % np.uint64(2 ** 61 - 1))


def gen_fullsize(variant, fname, pair, num_pairs, row_step=8):
    """BASELINE.json configs at their FULL size through the genuine reference (name-keyed synthetic weights, seed 7): per pair
    index the final outputs (coarse features complete, everything else as strided slices + float64 sums), for pair 0 also the
    output of every transformer layer (first call = ref direction) and the checksums of all pyramid index tables."""
    res = {'pair': np.array(pair), 'synth_seed': np.int64(7), 'num_pairs': np.int64(num_pairs), 'row_step': np.int64(row_step)}
    for index in range(num_pairs):
        cfg, model, (ref, src, T), dd, ops, feats, out = run_model(variant, micro=False, synth_seed=7, pair=pair, index=index,
                                                                    light=True)
        P = 'p%d/' % index
        res[P + 'lengths'] = np.stack([_np(l) for l in dd['lengths']])
        res[P + 'feats_c'] = _np(feats['c'])[::row_step, :, ::4] if index == 0 else _np(feats['c'])[::4 * row_step, :, ::8]
        res[P + 'feats_c_sum'] = np.float64(feats['c'].double().sum())
        res[P + 'feats_c_abs'] = np.float64(feats['c'].double().abs().max())
        res[P + 'feats_f'] = _np(feats['f'])[::4 * row_step] if index == 0 else _np(feats['f'])[::16 * row_step]
        res[P + 'feats_f_sum'] = np.float64(feats['f'].double().sum())
        res[P + 'feats_f_abs'] = np.float64(feats['f'].double().abs().max())
        for k in ('ref_feats_c', 'src_feats_c'):
            res[P + k] = _np(out[k]) if index == 0 else _np(out[k])[::row_step]
            res[P + k + '_sum'] = np.float64(out[k].double().sum())
        for k in ('ref_node_corr_indices', 'src_node_corr_indices', 'estimated_transform'):
            res[P + k] = _np(out[k])
        res[P + 'num_corr'] = np.int64(out['ref_corr_points'].shape[0])
        res[P + 'corr_score_sum'] = np.float64(out['corr_scores'].double().sum())
        # the dense correspondences themselves (points + scores): a test can then attribute every correspondence that differs to the threshold it
        # sits on (mutual top-k boundary, the 0.05 confidence) instead of only counting
        res[P + 'corr_ref_points'], res[P + 'corr_src_points'] = _np(out['ref_corr_points']), _np(out['src_corr_points'])
        res[P + 'corr_scores'] = _np(out['corr_scores'])
        ms = out['matching_scores']
        res[P + 'matching_scores_head'] = _np(ms[:4])
        res[P + 'matching_scores_rowsum'] = _np(ms[:, :-1, :-1].exp().sum((1, 2)))
        if index == 0:
            res['blocks'] = np.array(cfg.geotransformer.blocks)
            for name, rec in ops.items():
                t = torch.from_numpy(rec['out0'])
                res['op/%s/out0' % name] = rec['out0'][..., ::row_step, :]
                res['op/%s/sum' % name] = np.float64(t.double().sum())
                res['op/%s/abs' % name] = np.float64(t.double().abs().max())
            for key in ('neighbors', 'subsampling', 'upsampling'):
                res['checksum/' + key] = np.array([_index_checksum(t.numpy()) for t in dd[key]], dtype=np.uint64)
                res['width/' + key] = np.array([t.shape[1] for t in dd[key]])
                # the same with every row sorted first: invariant to the order inside groups of equal float32 distance, which
                # the reference leaves to an unstable std::sort
                res['rowset/' + key] = np.array([_index_checksum(np.sort(t.numpy(), 1)) for t in dd[key]], dtype=np.uint64)
            res['points_last'] = _np(dd['points'][-1])
        print(fname, 'pair', index, 'done', flush=True)
    np.savez_compressed(os.path.join(HERE, fname), **res)


def _table_geometry(dd):
    """(key, i) -> (query points, support points) of the ten index tables of a collated pyramid (geotransformer/utils/data.py:46-93)."""
    pts = [p.numpy() for p in dd['points']]
    geo = {}
    for i in range(len(pts)):
        geo['neighbors', i] = (pts[i], pts[i])
    for i in range(len(pts) - 1):
        geo['subsampling', i] = (pts[i + 1], pts[i])
        geo['upsampling', i] = (pts[i], pts[i + 1])
    return geo


def pack_table_checksums(dd, res, prefix=''):
    geo = _table_geometry(dd)
    for key in ('neighbors', 'subsampling', 'upsampling'):
        res[prefix + 'checksum/' + key] = np.array([_index_checksum(t.numpy()) for t in dd[key]], dtype=np.uint64)
        res[prefix + 'width/' + key] = np.array([t.shape[1] for t in dd[key]])
        res[prefix + 'rowset/' + key] = np.array([_index_checksum(np.sort(t.numpy(), 1)) for t in dd[key]], dtype=np.uint64)
        canon = [tie_canonical(*geo[key, i], t.numpy()) for i, t in enumerate(dd[key])]
        res[prefix + 'tiecanon/' + key] = np.array([_index_checksum(c[0]) for c in canon], dtype=np.uint64)
        res[prefix + 'tierows/' + key] = np.array([c[1] for c in canon])
        res[prefix + 'tieentries/' + key] = np.array([c[2] for c in canon])
    res[prefix + 'points_last'] = _np(dd['points'][-1])
    res[prefix + 'lengths'] = np.stack([_np(l) for l in dd['lengths']])


def gen_tiecanon():
    """Tie-canonical checksums (tie_canonical) of the reference's tables for the clouds that DO hold exact float32 distance ties: the
    KITTI-sized 20k+20k pair, the 30k+30k cap pair -- collate only, no model."""
    ref_shims.install()
    res = {}
    for name, stages, voxel, radius, limits in (('c3_20k', 5, 0.3, 1.275, [38, 36, 36, 38, 38]), ('cap_30k', 4, 0.025, 0.0625, [38, 36, 36, 38]),
                                                ('c2_5k', 4, 0.025, 0.0625, [38, 36, 36, 38])):
        ref, src, T = make_pair(name)
        dd = collate(ref, src, T, stages, voxel, radius, limits)
        pack_table_checksums(dd, res, name + '/')
        print(name, {k: res[name + '/tierows/' + k].tolist() for k in ('neighbors', 'subsampling', 'upsampling')}, flush=True)
    np.savez_compressed(os.path.join(HERE, 'table_tiecanon.npz'), **res)


def gen_demo(row_step=8):
    """The reference's only real data: data/demo/{ref,src,gt}.npy (18 977 + 15 953 points, experiments/se3ete.3dmatch/demo.py:44-58, neighbour
    limits [38, 36, 36, 38] at :53) through the genuine collate and SE3ET-E with the name-keyed synthetic weights (seed 7; the released
    checkpoint is not in the repository).  Stored: the three input arrays, stage lengths, checksums of all ten tables (order-sensitive, row
    sets, tie-canonical), the last stage's points, per-layer strided taps and the outputs, as gen_fullsize stores them for pair 0."""
    demo = '/root/reference/data/demo/'
    ref, src, T = (np.load(demo + f).astype(np.float32) for f in ('ref.npy', 'src.npy', 'gt.npy'))
    cfg, model, _, dd, ops, feats, out = run_model('se3ete.3dmatch', micro=False, synth_seed=7, pair=(ref, src, T), light=True)
    res = {'ref': ref, 'src': src, 'transform': T, 'synth_seed': np.int64(7), 'row_step': np.int64(row_step),
           'blocks': np.array(cfg.geotransformer.blocks)}
    pack_table_checksums(dd, res)
    P = 'p0/'
    res[P + 'lengths'] = res['lengths']
    res[P + 'feats_c'] = _np(feats['c'])[::row_step, :, ::4]
    res[P + 'feats_c_sum'] = np.float64(feats['c'].double().sum())
    res[P + 'feats_c_abs'] = np.float64(feats['c'].double().abs().max())
    res[P + 'feats_f'] = _np(feats['f'])[::4 * row_step]
    res[P + 'feats_f_sum'] = np.float64(feats['f'].double().sum())
    res[P + 'feats_f_abs'] = np.float64(feats['f'].double().abs().max())
    for k in ('ref_feats_c', 'src_feats_c'):
        res[P + k] = _np(out[k])
        res[P + k + '_sum'] = np.float64(out[k].double().sum())
    for k in ('ref_node_corr_indices', 'src_node_corr_indices', 'estimated_transform'):
        res[P + k] = _np(out[k])
    res[P + 'num_corr'] = np.int64(out['ref_corr_points'].shape[0])
    res[P + 'corr_score_sum'] = np.float64(out['corr_scores'].double().sum())
    ms = out['matching_scores']
    res[P + 'matching_scores_head'] = _np(ms[:4])
    res[P + 'matching_scores_rowsum'] = _np(ms[:, :-1, :-1].exp().sum((1, 2)))
    for name, rec in ops.items():
        t = torch.from_numpy(rec['out0'])
        res['op/%s/out0' % name] = rec['out0'][..., ::row_step, :]
        res['op/%s/sum' % name] = np.float64(t.double().sum())
        res['op/%s/abs' % name] = np.float64(t.double().abs().max())
    # Rows whose neighbour SET the reference's unstable sort decided (a group of exactly tied distances cut by the neighbour limit): against
    # the tables with ties in ascending index order (oracle.precompute == the HIP path, bit for bit) -- the rows and the reference's
    # content, so that a test can run the model on exactly the reference's neighbourhoods
    from oracle import se3et_oracle as O
    mine = O.precompute(torch.from_numpy(np.concatenate([ref, src], 0)), torch.tensor([len(ref), len(src)]), cfg.backbone.num_stages,
                        cfg.backbone.init_voxel_size, cfg.backbone.init_radius, [38, 36, 36, 38])
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for i, t in enumerate(dd[key]):
            a, b = t.numpy(), mine[key][i].numpy()
            differ = (np.sort(a, 1) != np.sort(b, 1)).any(1)
            if key == 'upsampling':
                differ |= a[:, 0] != b[:, 0]          # nearest_upsample reads column 0 (kpconv/functional.py:6-22): the ORDER of a tie matters there
            rows = np.nonzero(differ)[0]
            res['patch/%s_%d_rows' % (key, i)] = rows.astype(np.int32)
            res['patch/%s_%d_vals' % (key, i)] = a[rows].astype(np.int32)
            res['orderdiff/%s_%d' % (key, i)] = np.int64((a != b).any(1).sum())
    # the 3 nearest superpoints of every superpoint as the reference's own expression selects them (geotransformer.py:69-90: top-(k+1) of
    # the distance map, first column dropped) -- exact ties at the cut are torch.topk's choice (rows 179 / 344 of ref, 89 of src here)
    from geotransformer.modules.ops import pairwise_distance as ref_pairwise_distance
    n0 = int(dd['lengths'][-1][0])
    for name, pc in (('ref', dd['points'][-1][:n0]), ('src', dd['points'][-1][n0:])):
        dist = torch.sqrt(ref_pairwise_distance(pc.unsqueeze(0), pc.unsqueeze(0)))
        res['patch/knn3_' + name] = dist.topk(k=4, dim=2, largest=False)[1][0, :, 1:].numpy().astype(np.int32)
    print('demo pair: lengths', res['lengths'].tolist(), 'widths', res['width/neighbors'].tolist(), 'tie rows',
          {k: res['tierows/' + k].tolist() for k in ('neighbors', 'subsampling', 'upsampling')}, 'corr', int(res[P + 'num_corr']), flush=True)
    np.savez_compressed(os.path.join(HERE, 'demo_se3ete.npz'), **res)


def gen_precompute_cap():
    """The 2000-superpoint cap of the coarsest stage (geotransformer/utils/data.py:34-43) through the genuine collate."""
    ref_shims.install()
    ref, src, T = make_pair('cap_30k')
    dd = collate(ref, src, T, 4, 0.025, 0.0625, [38, 36, 36, 38])
    res = {'lengths': np.stack([_np(l) for l in dd['lengths']]), 'points_last': _np(dd['points'][-1])}
    for key in ('neighbors', 'subsampling', 'upsampling'):
        res['checksum/' + key] = np.array([_index_checksum(t.numpy()) for t in dd[key]], dtype=np.uint64)
        res['rowset/' + key] = np.array([_index_checksum(np.sort(t.numpy(), 1)) for t in dd[key]], dtype=np.uint64)
        res['width/' + key] = np.array([t.shape[1] for t in dd[key]])
    print('cap_30k stage lengths', res['lengths'].tolist())
    np.savez_compressed(os.path.join(HERE, 'precompute_cap.npz'), **res)


def gen_train(variant, fname, micro=True, pair='micro', synth_seed=None, full_grads=None):
    """BASELINE.json configs[4] pieces through the genuine reference in TRAINING mode: forward with ground-truth superpoint
    targets, OverallLoss (weighted circle loss + fine NLL), backward, one Adam step.  Stored: the loss values, the ground-truth
    correspondences, the gradient norm, a strided slice and the largest magnitude of every parameter's gradient, a few complete gradients, the
    total norm, and a parameter checksum
    after the optimizer step.  Micro config with the reference-initialised weights of micro_se3ete.npz (seed 0)."""
    make_cfg, create_model = ref_shims.load_experiment(variant)
    import loss as loss_mod                     # experiments/<variant>/loss.py
    cfg = make_cfg()
    if micro:
        cfg.backbone.init_dim, cfg.backbone.output_dim, cfg.backbone.group_norm = 8, 32, 4
        cfg.geotransformer.input_dim, cfg.geotransformer.hidden_dim, cfg.geotransformer.output_dim = 128, 32, 32
    torch.manual_seed(0)
    model = create_model(cfg).train()
    if synth_seed is not None:      # full-size models: name-keyed synthetic weights (regenerated on the other side, not stored)
        sd = model.state_dict()
        for k in sd:
            if k.rsplit('.', 1)[-1] in LEARNED_LEAVES and sd[k].dtype == torch.float32 and 'anchors' not in k:
                sd[k] = torch.from_numpy(np.asarray(synth_tensor(k, sd[k].shape, synth_seed)))
        model.load_state_dict(sd)
    loss_fn = loss_mod.OverallLoss(cfg)
    ref, src, T = make_pair(pair)
    dd = collate(ref, src, T, cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius,
                 [38, 36, 36, 38, 38][:cfg.backbone.num_stages])
    np.random.seed(0)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.optim.lr, weight_decay=cfg.optim.weight_decay)
    target = {}
    model.coarse_target.register_forward_hook(lambda m, a, o: target.update(ref=o[0], src=o[1], overlaps=o[2]))
    out = model(dd)
    losses = loss_fn(out, dd)
    opt.zero_grad()
    losses['loss'].backward()
    res = {'ref': ref, 'src': src, 'transform': T, 'lr': np.float64(cfg.optim.lr), 'weight_decay': np.float64(cfg.optim.weight_decay)}
    for k in ('loss', 'c_loss', 'f_loss'):
        res['loss/' + k] = np.float64(losses[k].detach())
    res['gt_node_corr_indices'] = _np(out['gt_node_corr_indices'])
    res['target/ref'], res['target/src'], res['target/overlaps'] = _np(target['ref']), _np(target['src']), _np(target['overlaps'])
    res['gt_node_corr_overlaps'] = _np(out['gt_node_corr_overlaps'])
    res['ref_node_corr_knn_masks_sum'] = np.int64(out['ref_node_corr_knn_masks'].sum())
    res['matching_scores_shape'] = np.array(out['matching_scores'].shape)
    names, norms = [], []
    total = 0.0
    for n, p_ in model.named_parameters():
        if p_.grad is None:
            continue
        names.append(n)
        g = p_.grad.double()
        norms.append(float(g.norm()))
        total += float((g * g).sum())
    res['grad/names'], res['grad/norms'], res['grad/total_norm'] = np.array(names), np.array(norms), np.float64(total ** 0.5)
    # a strided slice (up to 64 entries) and the largest magnitude of EVERY gradient: a wrong gradient with the right norm does not pass
    slices, offsets, absmax = [], [0], []
    params_ = dict(model.named_parameters())
    for n in names:
        flat = params_[n].grad.reshape(-1)
        sl = flat[::max(1, flat.numel() // 64)][:64]
        slices.append(_np(sl))
        offsets.append(offsets[-1] + sl.numel())
        absmax.append(float(flat.abs().max()))
    res['grad/slices'], res['grad/slice_offsets'], res['grad/absmax'] = np.concatenate(slices), np.array(offsets), np.array(absmax)
    for n in full_grads or ('backbone.encoder1_1.interso3.conv.weights', 'backbone.encoder4_3.unary2.mlp.weight',
                            'transformer.embedding.proj_d.weight', 'transformer.transformer.layers.0.attention.attention.proj_eq.weight',
                            'transformer.transformer.layers.3.attention.attention.proj_q.weight', 'transformer.out_proj.weight',
                            'optimal_transport.alpha', 'transformer.transformer.rotcompress.expand.weight'):
        if n in dict(model.named_parameters()) and dict(model.named_parameters())[n].grad is not None:
            res['grad/full/' + n] = _np(dict(model.named_parameters())[n].grad)
    opt.step()
    res['after_step/param_sum'] = np.float64(sum(float(p_.detach().double().sum()) for p_ in model.parameters()))
    res['after_step/out_proj_weight'] = _np(model.transformer.out_proj.weight)
    print(fname, 'loss', float(losses['loss']), 'c', float(losses['c_loss']), 'f', float(losses['f_loss']), 'gt corr',
          out['gt_node_corr_indices'].shape[0], 'total grad norm', total ** 0.5)
    np.savez_compressed(os.path.join(HERE, fname), **res)


def gen_vgtk():
    """Inputs / outputs of the importable PyTorch twins of the EPN toolkit's CUDA kernels (vgtk/spconv/functional.py:373-399
    inter_zpconv_grouping_naive, :252-270 intra_zpconv_grouping_naive, batched_index_select) on seeded random data, incl. the gradient of the inter grouping."""
    ref_shims.install()
    import vgtk.spconv.functional as L
    g = torch.Generator().manual_seed(5)
    b, p, q, a, ks, nn_, c = 2, 37, 53, 6, 5, 9, 7
    idx = torch.randint(0, q, (b, p, nn_), generator=g)
    w = torch.rand(b, p, a, ks, nn_, generator=g)
    feats = torch.randn(b, c, q, a, generator=g, requires_grad=True)
    out = L.inter_zpconv_grouping_naive(idx, w, feats)                       # (b, c, ks, p, a)
    cot = torch.randn(out.shape, generator=g)
    (out * cot).sum().backward()
    pts = torch.randn(b, c, q, generator=g)
    gi = torch.randint(0, q, (b, 41), generator=g)
    gathered = L.batched_index_select(pts, 2, gi)
    # intra (anchor-axis) grouping: vgtk/spconv/functional.py:252-270 intra_zpconv_grouping_naive, forward + gradient (drawn AFTER the
    # entries above, so those keep their values)
    na_in, na_out, iks, ann, ip, ic = 12, 12, 3, 4, 31, 5
    intra_idx = torch.randint(0, na_in, (na_out, ann), generator=g)
    intra_w = torch.rand(na_out, iks, ann, generator=g)
    intra_feats = torch.randn(b, ic, ip, na_in, generator=g, requires_grad=True)
    intra_out = L.intra_zpconv_grouping_naive(intra_idx, intra_w, intra_feats)           # (b, c, ks, p, na_out)
    intra_cot = torch.randn(intra_out.shape, generator=g)
    (intra_out * intra_cot).sum().backward()
    np.savez_compressed(os.path.join(HERE, 'vgtk_ops.npz'), inter_idx=idx.numpy().astype(np.int32), inter_w=_np(w), feats=_np(feats),
                        inter_out=_np(out), cotangent=_np(cot), feats_grad=_np(feats.grad), points=_np(pts),
                        gather_idx=gi.numpy().astype(np.int32), gathered=_np(gathered),
                        intra_idx=intra_idx.numpy().astype(np.int32), intra_w=_np(intra_w), intra_feats=_np(intra_feats),
                        intra_out=_np(intra_out), intra_cotangent=_np(intra_cot), intra_feats_grad=_np(intra_feats.grad))


if __name__ == '__main__':
    which = sys.argv[1:] or ['tables', 'precompute', 'micro', 'synthw']
    if 'tables' in which:
        gen_tables()
    if 'precompute' in which:
        gen_precompute()
    if 'micro' in which:
        gen_micro('se3ete.3dmatch', 'micro_se3ete.npz')
        gen_micro('se3eti.3dmatch', 'micro_se3eti.npz')
    if 'synthw' in which:
        gen_synthw('se3ete2.3dmatch', 'synthw_se3ete2.npz')
        gen_synthw('se3eti2.3dmatch', 'synthw_se3eti2.npz')
        gen_synthw('se3ete.3dmatch', 'synthw_se3ete.npz')
    if 'c1' in which or 'synthw' in which:
        # BASELINE.json configs[0]: SE3ET-I2 on the 2k+2k pair (outputs only)
        gen_synthw('se3eti2.3dmatch', 'synthw_se3eti2_c1.npz', pair='c1_2k', keep_ops=False)
    if 'kitti' in which or 'synthw' in which:
        gen_synthw('se3eti.kitti', 'synthw_se3eti_kitti.npz', pair='c3_4k')
    if 'vgtk' in which:
        gen_vgtk()
    if 'train' in which:
        gen_train('se3ete.3dmatch', 'train_micro_se3ete.npz')
        gen_train('se3eti.3dmatch', 'train_micro_se3eti.npz')
    if 'train_fullsize' in which:
        # BASELINE.json configs[4] at its own size: SE3ET-E, the 5k+5k pair 0 of the bench workload, synthetic weights (seed 7)
        gen_train('se3ete.3dmatch', 'train_c2_se3ete_5k.npz', micro=False, pair='c2_5k', synth_seed=7,
                  full_grads=('backbone.encoder1_1.interso3.conv.weights', 'backbone.encoder2_2.interso3.conv.weights',
                              'transformer.embedding.proj_d.weight', 'transformer.transformer.layers.0.attention.attention.proj_eq.weight',
                              'transformer.transformer.layers.3.attention.attention.proj_q.weight', 'transformer.out_proj.weight',
                              'optimal_transport.alpha'))
    if 'cap' in which:
        gen_precompute_cap()
    if 'tiecanon' in which:
        gen_tiecanon()
    if 'demo' in which:
        gen_demo()
    if 'fullsize' in which:
        gen_fullsize('se3ete.3dmatch', 'c2_se3ete_5k.npz', 'c2_5k', 8)
    if 'fullsize_kitti' in which or 'fullsize' in which:
        gen_fullsize('se3eti.kitti', 'c3_se3eti_kitti_20k.npz', 'c3_20k', 1)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
