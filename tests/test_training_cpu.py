"""CPU: the plain-torch pieces of the training step against the reference's training fixture, and the data-parallel wiring on
world_size 2 over gloo (the gradient all-reduce of BASELINE.json configs[4]; RCCL on the GPUs)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from helpers import assert_close


def test_ground_truth_correspondences_and_coarse_loss_match_reference(golden_dir):
    """se3et_amd.training.node_correspondences / OverallLoss.coarse on the oracle's forward outputs (features pinned to the reference at
    1e-5) against gt_node_corr_* and c_loss captured from the genuine reference's training step."""
    from oracle import se3et_oracle as O
    from se3et_amd.model import make_cfg
    from se3et_amd.training import OverallLoss, node_correspondences, select_targets
    g = np.load(golden_dir + '/train_micro_se3ete.npz')
    w = np.load(golden_dir + '/micro_se3ete.npz')
    sd = {k[3:]: torch.from_numpy(w[k]) for k in w.files if k.startswith('sd/')}
    cfg = make_cfg('micro_e')
    oc = O.OracleConfig.from_model_cfg(cfg)
    pts = torch.from_numpy(np.concatenate([g['ref'], g['src']], 0))
    b = cfg.backbone
    data = O.precompute(pts, torch.tensor([len(g['ref']), len(g['src'])]), b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    data['features'] = torch.ones((pts.shape[0], 1))
    with torch.no_grad():
        out = O.forward(sd, oc, data, with_lgr=False)
    n_c, n_f = int(data['lengths'][-1][0]), int(data['lengths'][1][0])
    pc, pf = data['points'][-1], data['points'][1]
    T = torch.from_numpy(g['transform'])
    parts = []
    for nodes, fine in ((pc[:n_c], pf[:n_f]), (pc[n_c:], pf[n_f:])):
        _, nm, knn, km = O.point_to_node_partition(fine, nodes, cfg.model.num_points_in_patch)
        parts.append((nodes, torch.cat((fine, torch.zeros(1, 3)))[knn], nm, km))
    (rn, rk, rnm, rkm), (sn, sk, snm, skm) = parts
    gi, go = node_correspondences(rn, sn, rk, sk, T, cfg.model.ground_truth_matching_radius, rnm, snm, rkm, skm)
    want = {(int(a), int(b_)): float(o) for (a, b_), o in zip(g['gt_node_corr_indices'], g['gt_node_corr_overlaps'])}
    got = {(int(a), int(b_)): float(o) for (a, b_), o in zip(gi.tolist(), go.tolist())}
    assert set(got) == set(want) and max(abs(got[k] - want[k]) for k in want) < 1e-6
    loss = OverallLoss(cfg).coarse({'ref_feats_c': out['ref_feats_c'], 'src_feats_c': out['src_feats_c'],
                                    'gt_node_corr_indices': gi, 'gt_node_corr_overlaps': go})
    assert abs(float(loss) - float(g['loss/c_loss'])) <= 1e-4 * float(g['loss/c_loss'])
    # target selection: the reference drew with numpy's global generator seeded 0
    np.random.seed(0)
    tr, ts, to = select_targets(gi, go, cfg.coarse_matching.num_targets, cfg.coarse_matching.overlap_threshold)
    assert sorted(zip(tr.tolist(), ts.tolist())) == sorted(zip(g['target/ref'].tolist(), g['target/src'].tolist()))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(torch.nn.Module):
    """Stand-in with the shape of the problem (two feature sets -> circle loss); the SE3ET forward itself needs the GPU kernels."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a, self.b = torch.nn.Linear(6, 8), torch.nn.Linear(6, 8)
        self.unused = torch.nn.Parameter(torch.ones(3))            # never reached by the loss, as the rotation heads of the KITTI model

    def forward(self, x, y):
        return torch.nn.functional.normalize(self.a(x), dim=1), torch.nn.functional.normalize(self.b(y), dim=1)


def _toy_loss(model, seed):
    from se3et_amd.model import make_cfg
    from se3et_amd.training import OverallLoss
    g = torch.Generator().manual_seed(seed)
    x, y = torch.randn(12, 6, generator=g), torch.randn(10, 6, generator=g)
    r, s = model(x, y)
    gi = torch.tensor([[0, 1], [3, 2], [5, 5], [7, 9]])
    return OverallLoss(make_cfg('micro_e')).coarse({'ref_feats_c': r, 'src_feats_c': s, 'gt_node_corr_indices': gi,
                                                    'gt_node_corr_overlaps': torch.tensor([0.5, 0.3, 0.9, 0.2])})


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from se3et_amd import sharding
    from se3et_amd.model import make_cfg
    from se3et_amd.training import distributed_model, make_optimizer
    sharding.init_distributed('gloo')
    model = _Toy()
    ddp = distributed_model(model)
    opt = make_optimizer(ddp, make_cfg('micro_e'), world_size=world)
    loss = _toy_loss(ddp, seed=100 + rank)                         # every rank its own pair
    opt.zero_grad()
    loss.backward()
    grads = {n: p.grad.numpy().copy() for n, p in model.named_parameters() if p.grad is not None}     # numpy: plain pickles
    opt.step()
    q.put((rank, float(loss.detach()), grads, opt.param_groups[0]['lr'], model.a.weight.detach().numpy().copy()))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_all_reduce():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(30)
    # single-process reference: gradient of the mean of the two per-rank losses
    model = _Toy()
    total = (_toy_loss(model, 100) + _toy_loss(model, 101)) / 2
    total.backward()
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        for r in range(2):
            assert_close(res[r][2][n], p.grad, 1e-5, 'rank %d grad %s' % (r, n))
    assert res[0][3] == res[1][3] == 2e-4                          # lr x world size (engine/base_trainer.py:191-196)
    assert np.array_equal(res[0][4], res[1][4])                     # replicas stay identical after the step
    assert abs(res[0][1] - res[1][1]) > 1e-6                        # they did see different pairs
