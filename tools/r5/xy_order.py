import numpy as np, torch, itertools
g=np.load('tests/golden/demo_se3ete.npz')
P=torch.from_numpy(g['points_last'])[:410].contiguous()
xy=torch.matmul(P.unsqueeze(0), P.unsqueeze(0).transpose(-1,-2))[0].numpy()
p=P.numpy(); a=p[:,None,:].astype(np.float64); b=p[None,:,:].astype(np.float64)
f32=lambda x: x.astype(np.float32)
def fma(x,y,c): return f32(x.astype(np.float64)*y.astype(np.float64)+c.astype(np.float64))
A=p[:,None,:]; B=p[None,:,:]
cands={}
for perm in itertools.permutations(range(3)):
    i,j,k=perm
    # plain mul/add
    cands['plain%s'%(perm,)]=f32(f32(f32(A[...,i]*B[...,i])+f32(A[...,j]*B[...,j]))+f32(A[...,k]*B[...,k]))
    cands['fma%s'%(perm,)]=fma(A[...,k],B[...,k],fma(A[...,j],B[...,j],f32(A[...,i]*B[...,i])))
    cands['fma0%s'%(perm,)]=fma(A[...,k],B[...,k],fma(A[...,j],B[...,j],fma(A[...,i],B[...,i],np.zeros_like(xy))))
for n,c in cands.items():
    print(n, 'mismatch', int((c!=xy).sum()), 'of', xy.size)
x2=(P**2).sum(-1).numpy()
for perm in itertools.permutations(range(3)):
    i,j,k=perm
    c=f32(f32(f32(p[:,i]*p[:,i])+f32(p[:,j]*p[:,j]))+f32(p[:,k]*p[:,k]))
    print('x2 plain',perm,int((c!=x2).sum()))
import subprocess
print(subprocess.run('lscpu | grep -E "Model name|Flags" | cut -c1-300', shell=True, capture_output=True, text=True).stdout)
print(torch.__config__.show()[:600])
sq = (torch.from_numpy(x2)[:, None] - 2 * torch.from_numpy(xy) + torch.from_numpy(x2)[None, :]).clamp(min=0)
from_ref = torch.sqrt(sq).diag()
print('nonzero self distances (torch on this CPU):', int((from_ref != 0).sum()), from_ref[:40].tolist())
