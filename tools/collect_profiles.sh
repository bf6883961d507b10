#!/bin/bash
# Everything under profiles/rNN_* in one GPU call: tools/collect_profiles.sh r03   (writes gpurun_out/profiles_r03/; copy what is to be kept)
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-rXX}; O=$R/gpurun_out/profiles_$tag; mkdir -p $O; cd $R
timeout 900 python bench.py > $O/${tag}_bench.json 2> $O/bench.err
# the driver's exact command (BENCH_rNN.json), twice
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench_20_5.json 2>> $O/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${tag}_bench_20_5_b.json 2>> $O/bench.err
timeout 900 python bench.py --batch 1 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 > $O/${tag}_bench_batch1.json 2>> $O/bench.err
timeout 900 python bench.py --inflight 1 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 > $O/${tag}_bench_inflight1.json 2>> $O/bench.err
timeout 900 python bench.py --inflight 1 --prefetch 0 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 > $O/${tag}_bench_noprefetch.json 2>> $O/bench.err
timeout 900 python bench.py --variant se3eti_kitti --pair c3_20k --attention-dtype bfloat16 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 > $O/${tag}_bench_c3_bf16.json 2>> $O/bench.err
timeout 900 python bench.py --variant se3eti_kitti --pair c3_20k --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 > $O/${tag}_bench_c3_f32.json 2>> $O/bench.err
tools/prof.sh ${tag} bench.py --no-cpu-baseline --single-pair-steps 0 --train-steps 0 --roofline-quiet-steps 0 > $O/prof.txt 2>&1; cp gpurun_out/${tag}_kernel_stats.csv $O/${tag}_kernel_stats.csv
# one batch in flight, pyramid and model back to back: every kernel alone on the GPU (the per-step categories and the kernels' own durations)
tools/prof.sh ${tag}seq bench.py --inflight 1 --prefetch 0 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 --roofline-quiet-steps 0 >> $O/prof.txt 2>&1; cp gpurun_out/${tag}seq_kernel_stats.csv $O/${tag}_kernel_stats_sequential.csv
tools/prof.sh ${tag}b1 bench.py --batch 1 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 >> $O/prof.txt 2>&1; cp gpurun_out/${tag}b1_kernel_stats.csv $O/${tag}_kernel_stats_batch1.csv
python tools/step_breakdown.py $O/${tag}_kernel_stats_sequential.csv 189 > $O/${tag}_step_breakdown.txt 2>&1      # 9 warm-up + 3 regions of 60 steps (16 pairs each since round 6)
# BASELINE.json configs[2] (SE3ET-I KITTI configuration, 20k+20k pairs): every kernel alone, per-step categories (10 steps + 3 warm-up)
tools/prof.sh ${tag}c3 bench.py --variant se3eti_kitti --pair c3_20k --batch 4 --steps 10 --warmup 3 --inflight 1 --prefetch 0 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 --roofline-quiet-steps 0 >> $O/prof.txt 2>&1; cp gpurun_out/${tag}c3_kernel_stats.csv $O/${tag}_kernel_stats_c3.csv
python tools/step_breakdown.py $O/${tag}_kernel_stats_c3.csv 33 > $O/${tag}_step_breakdown_c3.txt 2>&1      # 3 warm-up + 3 regions of 10 steps
# the training step, kernel by kernel (5 steps + 2 warm-up)
tools/prof.sh ${tag}train tools/train_bench.py --steps 5 --warmup 2 >> $O/prof.txt 2>&1; cp gpurun_out/${tag}train_kernel_stats.csv $O/${tag}_kernel_stats_train.csv
# how busy the GPU is with three batches in flight
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --single-pair-steps 0 --train-steps 0 --roofline-quiet-steps 0 > $O/kt.log 2>&1; f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_busy.py $f > $O/${tag}_gpu_busy.txt 2>&1 )
timeout 300 python tools/micro/block_tail.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_block_tail.txt
timeout 300 python tools/micro/linear_stream_shapes.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_linear_stream_shapes.txt
timeout 300 python tools/micro/rpe_eq_breakdown.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_rpe_eq_breakdown.txt
timeout 300 python tools/micro/kpconv_paths.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_kpconv_paths.txt
timeout 300 python tools/micro/dense_norm_shapes.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_dense_norm_shapes.txt
for L in 2 5 8; do echo "== KPConv layer $L (tools/micro/kpconv_layer.py $L fused): counters of kpconv_fused_kernel, average per dispatch"; tools/pmc_kernel.sh kpconv_fused_kernel tools/micro/kpconv_layer.py $L fused 3; tail -1 gpurun_out/pmck.log | grep layer; done > $O/${tag}_pmc_kpconv.txt 2>&1
for L in 0 2 5; do echo "== KPConv layer $L (tools/micro/kpconv_layer.py $L union): counters of kpconv_union_kernel, average per dispatch"; tools/pmc_kernel.sh kpconv_union_kernel tools/micro/kpconv_layer.py $L union 3; tail -1 gpurun_out/pmck.log | grep layer; done > $O/${tag}_pmc_kpconv_union.txt 2>&1
timeout 300 python tools/r5/union_check.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_kpconv_union_check.txt
timeout 300 python tools/r5/union_variants.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_kpconv_union_variants.txt
tools/pmc_passes.sh > $O/${tag}_pmc_attention_raw.txt 2>&1
tools/pmc_step.sh ${tag} > $O/pmc_step.log 2>&1; cp gpurun_out/${tag}_pmc_step.txt $O/${tag}_pmc_step.txt
for K in rpe_bias_kernel attention_x6_kernel; do echo "== $K (tools/pmc_attention.py: 16 clouds per launch; equivariant and invariant dispatches averaged together)"; tools/pmc_kernel.sh $K tools/pmc_attention.py; done > $O/${tag}_pmc_attention_sq.txt 2>&1
timeout 600 python tools/train_bench.py --steps 5 --warmup 2 --profile > $O/${tag}_train_bench.txt 2>&1
timeout 300 python tools/r6/tie_cost.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_tie_cost.txt
tools/prof.sh ${tag}demo tools/r6/demo_pyramid.py 8 >> $O/prof.txt 2>&1 < /dev/null; cp gpurun_out/${tag}demo_kernel_stats.csv $O/${tag}_kernel_stats_demo_pyramid.csv; grep pyramid gpurun_out/prof_${tag}demo.log > $O/${tag}_demo_pyramid_times.txt
timeout 300 python tools/micro/sinkhorn_ab.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/${tag}_sinkhorn_ab.txt
for V in true false; do echo "== sinkhorn_kernel<8, 9, $V> (tools/micro/sinkhorn_ab.py one: 2 048 patch pairs of 64 x 64, 100 iterations; true = base 2 with carried shifts, false = the reference's order of operations), average per dispatch"; tools/pmc_kernel.sh "sinkhorn_kernel<8, 9, $V>" tools/micro/sinkhorn_ab.py one; done > $O/${tag}_pmc_sinkhorn.txt 2>&1
ls -la $O
