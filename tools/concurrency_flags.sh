mkdir -p gpurun_out/r4c
run() { tag=$1; flags=$2; n=0; for i in 1 2 3 4 5 6 7 8 9 10 11 12; do PROBE_FLAGS=$flags PROBE_SAME=1 PROBE_FRESH=1 python tools/concurrency_probe.py 2 2,3 2 2>&1 | grep -q "rounds with differences: 0" || n=$((n+1)); done; echo "$tag: $n of 12 processes with differences" >> gpurun_out/r4c/flags.log; }
run base ""
run no_gram GRAM_KERNEL=0
run no_attn_f16 ATTENTION_F16=0
run no_x6 CROSS_EQ_BF16X6=0
run no_pending PENDING_NORM=0
run no_kpsplit KPCONV_SPLIT=0
cat gpurun_out/r4c/flags.log
