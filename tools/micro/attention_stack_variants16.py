"""Kernel variants of the stack-mode RPE self-attention call at the bench shape (16 clouds per launch); not a test."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib.util, time, torch
spec = importlib.util.spec_from_file_location('bas', 'tools/bench_attention_stack.py'); bas = importlib.util.module_from_spec(spec); spec.loader.exec_module(bas)
from se3et_amd._lib import lib
x = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < 1.5: y = x @ x
torch.cuda.synchronize()
lengths = (382, 350) * 8
for A, eq in ((6, True), (1, False)):
    for bv, sp in ((0, 0), (0, 2), (0, 4), (2, 2), (2, 3)):
        tb = min(bas.run(A, lengths, eq, bv, sp, 0, iters=10)[0] for _ in range(3))
        print('A=%d eq=%d bias variant %d split %d: %7.1f us' % (A, eq, bv, sp, tb), flush=True)
    for av in (0, 1, 2, 3):
        ta = min(bas.run(A, lengths, eq, 0, 0, av, iters=10)[1] for _ in range(3))
        print('A=%d eq=%d attention variant %d: %7.1f us' % (A, eq, av, ta), flush=True)
lib().se3_debug_set_bias_variant(0, 0); lib().se3_debug_set_attention_variant(0)
