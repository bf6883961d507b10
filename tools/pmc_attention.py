"""Runs the stack-mode RPE self-attention kernels a few times at the bench shape (8 pairs = 16 clouds per launch, N = 382 / 350,
C = 256, H = 4; equivariant A = 6 and invariant A = 1) for the rocprofv3 --pmc passes:
    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d out_fetch -- python3 tools/pmc_attention.py
    rocprofv3 --pmc WRITE_SIZE  --output-format csv -d out_write -- python3 tools/pmc_attention.py
and prints the algorithmic bytes of one call of each kind."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib.util
import torch
spec = importlib.util.spec_from_file_location('bas', 'tools/bench_attention_stack.py'); bas = importlib.util.module_from_spec(spec); spec.loader.exec_module(bas)
C, H = 256, 4
lengths = (382, 350) * 8
for A, eq in ((6, True), (1, False)):
    proj, vt, embs, eqs, starts = bas.setup(A, lengths, eq)
    out = torch.zeros(A, proj.shape[1], C, device='cuda')
    for _ in range(4):
        bas.call(proj, vt, embs, eqs, starts, lengths, out)
    nbytes = sum(4 * (4 * A * n * C + n * n * C + (A * n * n * 4 if eq else 0)) for n in lengths)
    print('A=%d eq=%d: algorithmic bytes per call %d' % (A, eq, nbytes))
torch.cuda.synchronize()
