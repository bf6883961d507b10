mkdir -p gpurun_out/r4e
tools/prof.sh r04c3 bench.py --variant se3eti_kitti --pair c3_20k --batch 4 --steps 10 --warmup 3 --inflight 1 --prefetch 0 --no-cpu-baseline --single-pair-steps 0 --train-steps 0 --roofline-quiet-steps 0 > gpurun_out/r4e/prof_c3.txt 2>&1
cp gpurun_out/r04c3_kernel_stats.csv gpurun_out/r4e/r04_kernel_stats_c3.csv
python tools/step_breakdown.py gpurun_out/r4e/r04_kernel_stats_c3.csv 13 > gpurun_out/r4e/r04_step_breakdown_c3.txt
cat gpurun_out/r4e/r04_step_breakdown_c3.txt
head -25 gpurun_out/r4e/r04_kernel_stats_c3.csv | cut -c1-150
# training step: kernel-level profile of fwd+bwd
tools/prof.sh r04train tools/train_bench.py --steps 5 --warmup 2 > gpurun_out/r4e/prof_train.txt 2>&1
cp gpurun_out/r04train_kernel_stats.csv gpurun_out/r4e/r04_kernel_stats_train.csv
head -40 gpurun_out/r4e/r04_kernel_stats_train.csv | cut -c1-130
python tools/train_bench.py --steps 5 --warmup 2 --profile 2>&1 | grep -v amdgpu | tail -20
