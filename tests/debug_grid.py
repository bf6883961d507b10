import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import native
from se3et_amd import ops
def clouds(n1,n2,scale,seed):
    g=np.random.default_rng(seed); pts=(g.uniform(0,1,(n1+n2,3))*scale).astype(np.float32)
    return torch.from_numpy(pts), torch.tensor([n1,n2])
for (n1,n2,scale,voxel) in [(10,0,1.0,0.5),(40,0,1.0,0.2),(200,0,1.0,0.1),(5000,4000,1.0,0.05)]:
    pts,lens=clouds(n1,n2,scale,0); nrm=torch.zeros_like(pts)
    sp,sl,_=native.grid_subsample(pts,lens,nrm,voxel)
    gp,_,gl=ops.grid_subsample(pts.cuda(),lens,None,voxel)
    gl=gl.cpu(); m=int(gl.sum()); gp=gp[:m].cpu()
    print('case',n1,n2,voxel,'counts',gl.tolist(),sl.tolist())
    a=[tuple(r) for r in gp.tolist()]; b=[tuple(r) for r in sp.tolist()]
    print(' same set', set(a)==set(b), 'n diff rows', sum(x!=y for x,y in zip(a,b)), 'of', len(a))
    if set(a)!=set(b): print('  only gpu', len(set(a)-set(b)), 'only cpu', len(set(b)-set(a)))
    first=[i for i,(x,y) in enumerate(zip(a,b)) if x!=y][:5]; print(' first diff positions', first)
    if n1<=40:
        ib={r:i for i,r in enumerate([tuple(x) for x in pts.tolist()])}
        print(' gpu order', [ib.get(r,-1) for r in a]); print(' cpu order', [ib.get(r,-1) for r in b])
