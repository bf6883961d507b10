#!/bin/bash
# SQ / TCC counters of one kernel (counter-only passes, one rocprofv3 run per counter set; FETCH_SIZE and WRITE_SIZE in passes of their own,
# values in KiB): tools/pmc_kernel.sh <kernel substring> <script.py> [args...]  ->  average per dispatch
R=${GRAFT_REPO_ROOT:-$(pwd)}; pat=$1; shift; script=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_WAVES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmck
  ( cd $R && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmck -o p -- python3 $script "$@" > $R/gpurun_out/pmck.log 2>&1 )
  f=$(find /tmp/pmck -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$pat" <<'PY'
import collections, csv, sys
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        v = agg.setdefault(r['Counter_Name'], [0, 0.0]); v[0] += 1; v[1] += float(r['Counter_Value'])
for c, (n, t) in agg.items(): print('%-28s %16.0f  (%d dispatches)' % (c, t / n, n))
PY
done
