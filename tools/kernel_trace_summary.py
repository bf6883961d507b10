"""Per-dispatch summary of a rocprofv3 --kernel-trace CSV: python tools/kernel_trace_summary.py <kernel_trace.csv> [substring ...]
Groups dispatches by (kernel, grid, workgroup) and prints calls / average / total duration."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
subs = sys.argv[2:]
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if subs and not any(s in n for s in subs):
        continue
    short = re.sub(r'\(anonymous namespace\)::', '', n)
    short = re.sub(r'^void ', '', short).split('(')[0][:70]
    key = (short, r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0.0
for k, v in agg.items():
    tot += sum(v)
    print('%-72s grid %9s x %3s wg %4s  calls %4d  avg %9.1f us  total %9.1f ms' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), sum(v) / 1e3))
print('total %.1f ms' % (tot / 1e3))
