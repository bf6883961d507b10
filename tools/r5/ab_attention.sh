#!/bin/bash
# Alternating bench runs of the working tree's library and an A/B library (tools/r5/build_ab.sh): tools/r5/ab_attention.sh [lib] [pairs]
L=${1:-tools/r5/ab/lib_attn_unscaled.so}; N=${2:-2}
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --single-pair-steps 0 --train-steps 0 2>/dev/null | tail -1 > gpurun_out/ab_new_$i.json
  SE3_LIB=$L python bench.py --no-cpu-baseline --single-pair-steps 0 --train-steps 0 2>/dev/null | tail -1 > gpurun_out/ab_old_$i.json
done
python - <<P
import json
for i in range(1, $N + 1):
    for n in ('new', 'old'):
        d = json.loads(open('gpurun_out/ab_%s_%d.json' % (n, i)).read()); r = d['roofline']; q = r.get('quiet') or {}
        print(n, i, d['value'], 'timed: frac', r.get('frac'), 'attention us', r.get('attention_kernel_avg_us'), '| quiet: frac', q.get('frac'), 'attention us', q.get('attention_kernel_avg_us'), 'eq call', q.get('eq_call_avg_us'))
P
