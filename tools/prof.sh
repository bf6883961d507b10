#!/bin/bash
# Kernel-time summary of one script under rocprofv3 (run on the GPU box from the repository root): tools/prof.sh <tag> <script.py> [args...]
# writes gpurun_out/<tag>_kernel_stats.csv and prints its first rows (name, calls, average ns).  The script runs with the repository root as
# its working directory (tools that add '.' to sys.path or open tools/... rely on it); only the profiler's scratch files go to /tmp.
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift; script=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $script "$@" > $R/gpurun_out/prof_$tag.log 2>&1 )
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${tag}_kernel_stats.csv; rm -rf $R/gpurun_out/prof_$tag
python3 - <<PY
import csv
rows = list(csv.reader(open('$R/gpurun_out/${tag}_kernel_stats.csv')))[1:9]
for r in rows: print('%-70s calls %5s avg %10.1f us  %5.1f%%' % (r[0][:70], r[1], float(r[3]) / 1e3, float(r[4])))
PY
