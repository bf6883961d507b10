"""GPU time per bench step by kernel family, from a rocprofv3 kernel_stats CSV of `bench.py --inflight 1 --prefetch 0 --no-cpu-baseline
--single-pair-steps 0 --train-steps 0` (every kernel alone on the GPU; 60 steps + 9 warm-up = 69):
python tools/step_breakdown.py <kernel_stats.csv> [steps incl. warm-up]"""
import csv, sys
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 69
fam = [('Cijk_', 'library GEMM'), ('linear_', 'dense f16-split GEMM'), ('dense_norm', 'dense + GroupNorm fused'), ('gn_chain_apply', 'GroupNorm apply (pending forms)'),
       ('gn_', 'GroupNorm'), ('kpconv_gather', 'KPConv gather (G form)'),
       ('kpconv_fused', 'KPConv fused'), ('kpconv_union_kernel', 'KPConv fused'), ('kpconv_union_plan', 'KPConv union plan + order'), ('point_order', 'KPConv union plan + order'), ('kpconv_neighbor_table', 'KPConv neighbour table'), ('kpconv_', 'KPConv other'), ('rpe_bias', 'RPE logits'), ('attention_kernel', 'attention'), ('attention_x6', 'attention'), ('attn_split', 'attention'), ('x6_split', 'attention / cross_eq operand split'), ('cross_eq', 'cross_eq'), ('gram_', 'cross_eq'),
       ('geo_', 'geo embedding'), ('embedding_table', 'geo embedding'), ('knn3', 'geo embedding'), ('sinkhorn', 'sinkhorn'),
       ('radius_', 'radius search'), ('grid_', 'grid subsample'), ('order_kernel', 'grid subsample'), ('neighbor_max', 'neighbor max'),
       ('add_ln', 'layer norm'), ('elementwise', 'torch elementwise'), ('at::native', 'torch other'), ('rocclr', 'copies / fills')]
tot = {}
rows = list(csv.reader(open(sys.argv[1])))[1:]
for r in rows:
    name, ns = r[0], float(r[2])
    if 'MT256x256x16' in name and float(r[3]) > 5e5: continue     # the clock ramp-up GEMM of bench.py (4096^3, ~1 ms per call)
    key = next((f for s, f in fam if s in name), 'other HIP kernels')
    tot[key] = tot.get(key, 0.0) + ns
s = sum(tot.values())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print('%-32s %7.2f ms/step  %5.1f %%' % (k, v / steps / 1e6, 100 * v / s))
print('%-32s %7.2f ms/step' % ('total', s / steps / 1e6))
