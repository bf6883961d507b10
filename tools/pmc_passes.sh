#!/bin/bash
# The two counter passes of profiles/rNN_pmc_attention.csv on the GPU box (counters only -- no trace domains beside --pmc):
#   tools/pmc_passes.sh   -> gpurun_out/pmc_fetch.txt, gpurun_out/pmc_write.txt (average per dispatch, KiB)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
echo "# rocprofv3 --pmc passes of tools/pmc_attention.py (tools/pmc_passes.sh), collected at commit ${SE3_COMMIT:-unknown}"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  ( cd $R && timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 tools/pmc_attention.py > $R/gpurun_out/pmc_$c.log 2>&1 )
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" > $R/gpurun_out/pmc_$c.txt <<'PY'
import collections, csv, sys
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'rpe_bias_kernel' in n or 'attention_kernel' in n or 'attention_x6_kernel' in n or 'attn_split_kv_kernel' in n or 'x6_split_kernel' in n:
        k = (n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0], r['Grid_Size'], r['Counter_Name'])
        v = agg.setdefault(k, [0, 0.0]); v[0] += 1; v[1] += float(r['Counter_Value'])
for (n, g, c), (cnt, tot) in agg.items():
    print('"%s",%s,%s,%d,%.1f' % (n, g, c, cnt, tot / cnt))
PY
  cat $R/gpurun_out/pmc_$c.txt
done
grep 'algorithmic bytes per call' $R/gpurun_out/pmc_WRITE_SIZE.log | sed 's/^/# /'
