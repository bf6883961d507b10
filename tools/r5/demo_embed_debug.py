"""Diagnostic (GPU): geometric / equivariant embedding of the demo pair's superpoints, HIP against the oracle, component by component."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import rel_err
from oracle import se3et_oracle as O
from se3et_amd import ops
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
g = np.load(os.path.join(ROOT, 'tests/golden/demo_se3ete.npz'))
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg), 7).cuda().eval()
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
oc = O.OracleConfig.from_model_cfg(cfg)
P = torch.from_numpy(g['points_last']); L = g['lengths'][-1]
for name, pts in (('ref', P[:L[0]]), ('src', P[L[0]:])):
    want = O.geometric_embedding(sd, 'transformer.embedding.', pts, oc)
    weq = O.equiv_embedding(sd, 'transformer.embedding.', pts)
    emb_mod = model.transformer.embedding
    got, geq = emb_mod(pts.cuda().unsqueeze(0))
    got, geq = got[0].cpu(), geq[0].cpu()
    err = (got - want).abs().amax(-1) / want.abs().max()
    n, m = divmod(int(err.argmax()), err.shape[1])
    d = torch.sqrt(O.pairwise_distance(pts, pts))
    print(name, 'emb rel err', rel_err(got, want), 'worst (n, m)', (n, m), 'dist', float(d[n, m]), 'entries > 1e-4:', int((err > 1e-4).sum()), 'of', err.numel())
    bad = torch.nonzero(err > 1e-4)
    print('  bad rows', sorted(set(bad[:, 0].tolist()))[:20], 'bad cols', sorted(set(bad[:, 1].tolist()))[:20])
    print('  eq rel err', rel_err(geq, weq))
    knn_w = d.topk(4, dim=1, largest=False)[1][:, 1:]
    knn_g = ops.knn3_stack(pts.cuda(), [len(pts)]).cpu()
    diff = (knn_w.sort(1)[0] != knn_g.sort(1)[0]).any(1)
    print('  knn rows differing (as sets)', int(diff.sum()), torch.nonzero(diff)[:, 0].tolist()[:10])
    for r in torch.nonzero(diff)[:, 0].tolist()[:4]:
        print('   row', r, 'oracle', knn_w[r].tolist(), d[r, knn_w[r]].tolist(), 'hip', knn_g[r].tolist(), d[r, knn_g[r]].tolist(), 'self d', float(d[r, r]))
    # entries outside the rows whose 3-NN set differs
    err2 = err.clone(); err2[diff] = 0
    bad = torch.nonzero(err2 > 1e-4)
    print('  bad entries outside knn-tie rows:', len(bad))
    d_idx, a_idx = O.embedding_indices(pts, oc.sigma_d, oc.sigma_a, oc.angle_k)
    for n, m in bad[:12].tolist():
        kn = knn_w[n]
        print('   (%d, %d) err %.2e d %.4f a_idx %s  knn dists %s  m in knn(n): %s' % (n, m, float(err2[n, m]), float(d[n, m]), [round(v, 4) for v in a_idx[n, m].tolist()],
              [round(v, 5) for v in d[n, kn].tolist()], m in kn.tolist()))
    # with the oracle's knn handed to the HIP kernel
    got2 = ops.geometric_embedding(pts.cuda(), emb_mod.embedding.div_term, emb_mod.proj_d.weight, emb_mod.proj_d.bias, emb_mod.proj_a.weight, emb_mod.proj_a.bias,
                                   emb_mod.sigma_d, emb_mod.sigma_a, 3, knn=knn_w.cuda().contiguous()).cpu()
    e3 = (got2 - want).abs().amax(-1) / want.abs().max()
    print('  with the oracle knn: rel err', rel_err(got2, want), 'entries > 1e-4', int((e3 > 1e-4).sum()), 'worst', divmod(int(e3.argmax()), e3.shape[1]))
