#!/bin/bash
# A/B of bench.py variants on ONE box, alternating runs: tools/ab_bench.sh <reps> "<env A>" "<env B>" [extra bench args]
# e.g. tools/ab_bench.sh 3 "SE3_LINEAR_STREAM=1" "SE3_LINEAR_STREAM=0"
reps=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out/ab
for i in $(seq 1 $reps); do
  for v in "$A" "$B"; do
    env $v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --single-pair-steps 0 --roofline-quiet-steps 0 "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab/ab.log
  done
done
