for i in 1 2; do for G in 0 2 3; do
  SE3_EQ_GROUPS=$G python bench.py --no-cpu-baseline --single-pair-steps 0 --train-steps 0 --roofline-quiet-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G=$G', d['value'], d['ms_per_step'])"
done; done
for G in 0 2 3; do SE3_EQ_GROUPS=$G tools/prof.sh eqg$G bench.py --inflight 1 --prefetch 0 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 --roofline-quiet-steps 0 --steps 20 --warmup 3 > /dev/null; echo "G=$G"; grep -h "cross_eq_apply_stack_x6\|dense_norm_kernel<2, 2, 1, 2, 4>\|dense_norm_kernel<2, 2, 1, 1, 4>" gpurun_out/eqg${G}_kernel_stats.csv | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:64].ljust(64), r[1], round(float(r[3])/1e3,1), 'us', round(float(r[2])/1e6/23,3), 'ms/step')
"; python3 tools/step_breakdown.py gpurun_out/eqg${G}_kernel_stats.csv 23 | grep -i "cross_eq\|dense\|total"; done
