#!/bin/bash
# HBM-side traffic of ONE bench step by kernel family: two counter passes (FETCH_SIZE, WRITE_SIZE: they do not fit
# one pass; counters only, no trace domain beside --pmc) over tools/one_step.py = ONE step of 8 pairs (a dispatch costs ~0.1 s under the
# counters: the bench command itself does not finish in 15 minutes).
#   tools/pmc_step.sh [tag]  -> gpurun_out/<tag>_pmc_step.txt   (copy to profiles/ to have bench.py's `roofline_step` read it)
# Per family: dispatches per step, FETCH_SIZE and WRITE_SIZE KiB per step as the counters report them, and the FETCH figure doubled for the
# families whose reads are 16-byte-per-lane streams (/opt/skills/guides/MI355X_MICROARCH.md "HBM": gfx950 tallies such 128-byte requests at
# 64 bytes; other access widths are uncalibrated and left as reported).
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-rXX}; STEPS=1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$c
  ( cd $R && timeout 1500 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcs_$c -o p -- python3 tools/one_step.py 1 > $R/gpurun_out/pmcs_$c.log 2>&1 )
done
python3 - "$(find /tmp/pmcs_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmcs_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $STEPS > $R/gpurun_out/${tag}_pmc_step.txt <<'PY'
import collections, csv, sys
fam = [('Cijk_', 'library GEMM'), ('dense_norm', 'dense + GroupNorm fused'), ('linear_', 'weight split'), ('gn_chain_apply', 'GroupNorm apply (pending forms)'),
       ('gn_', 'GroupNorm'), ('kpconv_gather', 'KPConv gather (G form)'), ('kpconv_fused', 'KPConv fused'), ('kpconv_neighbor_table', 'KPConv neighbour table'),
       ('kpconv_', 'KPConv other'), ('rpe_bias', 'RPE logits'), ('attention_kernel', 'attention'), ('attention_x6', 'attention'), ('attn_split', 'attention'),
       ('x6_split', 'attention / cross_eq operand split'), ('cross_eq', 'cross_eq'), ('gram_', 'cross_eq'), ('geo_', 'geo embedding'), ('embedding_table', 'geo embedding'),
       ('knn3', 'geo embedding'), ('sinkhorn', 'sinkhorn'), ('radius_', 'radius search'), ('grid_', 'grid subsample'), ('order_kernel', 'grid subsample'),
       ('neighbor_max', 'neighbor max'), ('add_ln', 'layer norm'), ('elementwise', 'torch elementwise'), ('at::native', 'torch other'), ('rocclr', 'copies / fills')]
# families whose loads are 16-byte-per-lane streams (float4 / buffer_load_b128 of contiguous rows): FETCH_SIZE x 2 (guide, gfx950)
wide = {'dense + GroupNorm fused', 'GroupNorm apply (pending forms)', 'GroupNorm', 'RPE logits', 'layer norm', 'sinkhorn'}
steps = int(sys.argv[3])
tot = {}
for i, path in enumerate(sys.argv[1:3]):
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        if 'MT256x256x16' in n: continue                       # the clock ramp-up GEMM of bench.py
        key = next((f for s, f in fam if s in n), 'other HIP kernels')
        t = tot.setdefault(key, [0, 0.0, 0.0])
        if i == 0: t[0] += 1
        t[1 + i] += float(r['Counter_Value'])
import os
print('# HBM-side traffic per bench step (8 pairs, one batch in flight, %d step incl. the one-off weight splits of a fresh process): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, KiB; collected at commit %s' % (steps, os.environ.get('SE3_COMMIT', 'unknown')))
print('# family, dispatches per step, FETCH_SIZE KiB per step (as reported), WRITE_SIZE KiB per step, FETCH corrected (x2 where the reads are 16 B / lane streams)')
sf = sw = sc = 0.0
for k, (n, f, w) in sorted(tot.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    c = f * (2 if k in wide else 1)
    sf += f; sw += w; sc += c
    print('"%s",%.1f,%.1f,%.1f,%.1f' % (k, n / steps, f / steps, w / steps, c / steps))
print('"total",,%.1f,%.1f,%.1f' % (sf / steps, sw / steps, sc / steps))
PY
cat $R/gpurun_out/${tag}_pmc_step.txt
