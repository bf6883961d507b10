"""What the equivariant RPE logits kernel pays over the invariant one, at the bench shape (16 clouds per launch): the kernel as it is,
without the equivariant term (same 24 folded-query rows, no eq-embedding reads), with the logits of all clouds written over one block
(1/16 of the written bytes), with other workgroup counts, and with the round-3 request order (bias variant 5).
python tools/micro/rpe_eq_breakdown.py"""
import os, sys; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
import time, torch
import bench_attention_stack as B
from se3et_amd._lib import lib
x = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < 1.5: y = x @ x
torch.cuda.synchronize()
lengths = (382, 350, 304, 310, 382, 350, 304, 310, 382, 350, 304, 310, 382, 350, 304, 310)
emb_gb = sum(n * n for n in lengths) * 256 * 4 / 1e9
for name, A, eq, bv in (('A=1', 1, False, 0), ('A=6', 6, True, 0)):
    B.run(A, lengths[:2], eq, bv, 0, 0, iters=1, check=True)
for rep in range(2):
    for name, A, eq, bv, split in (('A=6 eq (default)', 6, True, 0, 0), ('A=6 without the eq term', 6, False, 0, 0), ('A=6 eq, logits over one block', 6, True, 9, 0),
                                   ('A=6 no eq, logits over one block', 6, False, 9, 0), ('A=1 invariant', 1, False, 0, 0), ('A=1 invariant, logits over one block', 1, False, 9, 0),
                                   ('A=6 eq, 3 workgroups per CU queued', 6, True, 0, 3), ('A=6 eq, 4 workgroups per CU queued', 6, True, 0, 4),
                                   ('A=1 invariant, 3 workgroups per CU (136 registers: all resident)', 1, False, 0, 3),
                                   ('A=1 invariant, 4 workgroups per CU queued', 1, False, 0, 4),
                                   ('A=1 invariant, requests in operand order (round 3), 2 per CU', 1, False, 5, 2),
                                   ('A=6 eq, requests in operand order (round 3)', 6, True, 5, 0), ('A=6 no eq term, requests in operand order', 6, False, 5, 0)):
        tb, ta, nbytes = B.run(A, lengths, eq, bv, split, 0, iters=20)
        print('%-40s logits kernel %6.1f us  (embedding stream alone %.2f GB -> %.2f TB/s)   attention %6.1f us' % (name, tb, emb_gb, emb_gb / tb * 1e3, ta))
lib().se3_debug_set_bias_variant(0, 0)
