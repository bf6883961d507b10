import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import se3et_oracle as O
from se3et_amd import functional as SF
g = torch.Generator().manual_seed(8)
N, C = 20, 32
pts = torch.rand(N, 3, generator=g) * torch.tensor([1.5, 1.2, 1.0])
div = torch.exp(torch.arange(0, C, 2).float() * (-np.log(10000.0) / C))
st = {'e.embedding.div_term': div}
for n in ('d', 'a'):
    st['e.proj_%s.weight' % n] = torch.randn(C, C, generator=g) / C ** 0.5
    st['e.proj_%s.bias' % n] = torch.randn(C, generator=g) * 0.1
cfg = O.OracleConfig()
want = O.geometric_embedding(st, 'e.', pts, cfg)
c = lambda k: st[k].cuda()
got = SF.geometric_embedding(pts.cuda(), c('e.embedding.div_term'), c('e.proj_d.weight'), c('e.proj_d.bias'), c('e.proj_a.weight'), c('e.proj_a.bias'), cfg.sigma_d, cfg.sigma_a, 3).cpu()
err = (got - want).abs()
print('max err', err.max().item(), 'at', np.unravel_index(err.argmax().item(), err.shape))
print('per-n max', err.amax((1, 2))[:8])
print('per-m max', err.amax((0, 2))[:12])
print('per-c max', err.amax((0, 1))[:8])
# only distance part: zero the angle weights
z = torch.zeros_like(st['e.proj_a.weight']); zb = torch.zeros_like(st['e.proj_a.bias'])
st2 = dict(st); st2['e.proj_a.weight'] = z; st2['e.proj_a.bias'] = zb
w2 = O.geometric_embedding(st2, 'e.', pts, cfg)
g2 = SF.geometric_embedding(pts.cuda(), c('e.embedding.div_term'), c('e.proj_d.weight'), c('e.proj_d.bias'), z.cuda(), zb.cuda(), cfg.sigma_d, cfg.sigma_a, 3).cpu()
print('distance-only err', (g2 - w2).abs().max().item())
d_idx, a_idx = O.embedding_indices(pts, 0.2, 15.0, 3)
print('a_idx range', a_idx.min().item(), a_idx.max().item(), 'd range', d_idx.max().item())
