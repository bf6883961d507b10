"""Per-step timeline of the pairs-in-flight pipeline (not a test): python tools/pipeline_probe.py [P]"""
import sys, time, threading; sys.path.insert(0, '.')
import numpy as np, torch
from se3et_amd.data import precompute_data_stack_mode
from se3et_amd.model import create_model, load_synthetic_weights, make_cfg
from se3et_amd.synthetic import make_pair
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sys.setswitchinterval(float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3)
cfg = make_cfg('se3ete'); model = load_synthetic_weights(create_model(cfg)).cuda().eval()
b = cfg.backbone
N = 60
pairs = []
for s in range(N):
    ref, src, _ = make_pair('c2_5k', index=s)
    pairs.append((torch.from_numpy(np.concatenate([ref, src], 0)).cuda(), torch.tensor([len(ref), len(src)], dtype=torch.int64)))
feats = torch.ones((pairs[0][0].shape[0], 1), device='cuda')
def step(i):
    pts, lens = pairs[i]
    d = precompute_data_stack_mode(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
    d['features'] = feats
    return model(d)
x = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < 1.0: x @ x
_h = model.stage_hook; model.stage_hook = None
for i in range(12): step(i)
model.stage_hook = _h
torch.cuda.synchronize()
log = []
S1, S2 = threading.Lock(), threading.Lock()
STAGED = len(sys.argv) > 3 and sys.argv[3] == 'staged'
def hook():
    S2.acquire(); S1.release()
if STAGED: model.stage_hook = hook
def run(idx, stream, tid):
    with torch.cuda.stream(stream):
        for i in idx:
            t0 = time.perf_counter()
            if STAGED: S1.acquire()
            step(i)
            if STAGED: S2.release()
            t1 = time.perf_counter()
            log.append((tid, i, t0, t1))
        stream.synchronize()
streams = [torch.cuda.Stream() for _ in range(P)]
for rep in range(5):
    log.clear()
    idx = list(range(12, N))
    T0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(idx[t::P], streams[t], t)) for t in range(P)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    T1 = time.perf_counter()
    d = np.array([(t1 - t0) * 1e3 for _, _, t0, t1 in log])
    print('rep %d: P=%d  %.1f pairs/s; host step ms: mean %.2f min %.2f max %.2f p90 %.2f' % (rep, P, (N - 12) / (T1 - T0), d.mean(), d.min(), d.max(), np.percentile(d, 90)))
    print('   ', ' '.join('%.0f' % v for v in d[:48]))
