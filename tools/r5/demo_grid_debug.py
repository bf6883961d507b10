"""Diagnostic (GPU): HIP grid subsampling against the C oracle on the reference's real pair, stage by stage."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import native
from se3et_amd import ops
g = np.load(os.path.join(ROOT, 'tests/golden/demo_se3ete.npz'))
ref, src = g['ref'], g['src']
pts = torch.from_numpy(np.concatenate([ref, src], 0))
lengths = torch.tensor([len(ref), len(src)])
voxel = 0.025
for stage in range(3):
    voxel *= 2
    wp, wl, _ = native.grid_subsample(pts, lengths, torch.zeros_like(pts), voxel)
    sp, _, sl = ops.grid_subsample(pts.cuda(), lengths, None, voxel)
    sl = sl.cpu(); sp = sp[:int(sl.sum())].cpu()
    print('stage', stage + 1, 'voxel', voxel, 'lengths', sl.tolist(), wl.tolist(), 'equal', torch.equal(sp, wp))
    o = 0
    for c in range(2):
        a, b = sp[o:o + int(sl[c])], wp[o:o + int(wl[c])]
        o += int(wl[c])
        if a.shape != b.shape:
            print('  cloud', c, 'shape', a.shape, b.shape); continue
        sa = set(map(tuple, a.tolist())); sb = set(map(tuple, b.tolist()))
        neq = (a != b).any(1)
        first = int(torch.nonzero(neq)[0]) if neq.any() else -1
        print('  cloud', c, 'rows', len(a), 'same set', sa == sb, 'only mine', len(sa - sb), 'rows differing', int(neq.sum()), 'first', first)
        if first >= 0:
            # where does the oracle's row `first` sit in mine?
            idx = {tuple(r): i for i, r in enumerate(a.tolist())}
            print('   oracle rows', first, '..', first + 5, 'sit in mine at', [idx.get(tuple(r), -1) for r in b[first:first + 6].tolist()])
    pts, lengths = wp, wl
