"""Diagnostic (GPU): bisect the first transformer layer on the demo pair: in_proj, embeddings, RPE attention, epilogue -- HIP vs oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import rel_err
from oracle import se3et_oracle as O
from se3et_amd import functional as SF
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
g = np.load(os.path.join(ROOT, 'tests/golden/demo_se3ete.npz'))
cfg = make_cfg('se3ete')
model = load_synthetic_weights(create_model(cfg), 7).cuda().eval()
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
oc = O.OracleConfig.from_model_cfg(cfg)
gt = model.transformer
P = torch.from_numpy(g['points_last']); L = g['lengths'][-1]
pts = P[:L[0]]
N = len(pts)
gen = torch.Generator().manual_seed(3)
feats = torch.randn(N, 6, 1024, generator=gen) * 3.0
x_w = O._lin(sd, 'transformer.in_proj.', feats.transpose(0, 1))
x_g = SF.linear(feats.cuda(), gt.in_proj.weight, gt.in_proj.bias).transpose(0, 1).contiguous()
print('in_proj', rel_err(x_g.cpu(), x_w))
e_w = O.geometric_embedding(sd, 'transformer.embedding.', pts, oc); q_w = O.equiv_embedding(sd, 'transformer.embedding.', pts)
e_g, q_g = gt.embedding(pts.cuda().unsqueeze(0))
print('emb', rel_err(e_g[0].cpu(), e_w), 'eq', rel_err(q_g[0].cpu(), q_w))
lp = 'transformer.transformer.layers.0.'
layer = gt.transformer.layers[0]
with torch.no_grad():
    want, _ = O.rpe_layer(sd, lp, x_w, e_w, q_w, 4)
    hid_w, _ = O.rpe_attention(sd, lp + 'attention.attention.', x_w, x_w, e_w, q_w, 4)
    # HIP layer on the ORACLE's inputs
    out = layer(x_w.cuda().unsqueeze(0), x_w.cuda().unsqueeze(0), e_w.cuda().unsqueeze(0), equiv_states=q_w.cuda().unsqueeze(0))
    got = out[0][0].cpu()
    err = (got - want).abs().amax((0, 2)) / want.abs().max()
    print('layer0 on oracle inputs', rel_err(got, want), 'rows > 1e-4:', int((err > 1e-4).sum()), 'worst rows', err.topk(8)[1].tolist(), [float('%.2e' % v) for v in err.topk(8)[0]])
    hid_g = layer.attention.attention(x_w.cuda().unsqueeze(0), x_w.cuda().unsqueeze(0), x_w.cuda().unsqueeze(0), e_w.cuda().unsqueeze(0), embed_eq=q_w.cuda().unsqueeze(0))[0][0].cpu()
    eh = (hid_g - hid_w).abs().amax((0, 2)) / hid_w.abs().max()
    print('attention hidden', rel_err(hid_g, hid_w), 'rows > 1e-4:', int((eh > 1e-4).sum()), 'worst rows', eh.topk(8)[1].tolist(), [float('%.2e' % v) for v in eh.topk(8)[0]])
    # logits magnitude
    q = O._heads(O._lin(sd, lp + 'attention.attention.proj_q.', x_w), 4)
    print('|x|', float(x_w.abs().max()), '|q|', float(q.abs().max()), '|emb|', float(e_w.abs().max()))
