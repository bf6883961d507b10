"""How busy is the GPU in the bench's timed region?  From a rocprofv3 --kernel-trace CSV: the fraction of the window between the
dispatches at 30 % and 95 % of the dispatch count (the forwards; the clock ramp-up GEMMs and the set-up come before) during which at least
one kernel is running, and the average number of kernels running (sum of durations / window).
python tools/gpu_busy.py <kernel_trace.csv>"""
import csv, sys
iv = []
for r in csv.DictReader(open(sys.argv[1])):
    iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
iv.sort()
t0, t1 = iv[0][0], max(e for _, e in iv)
lo, hi = iv[int(0.30 * len(iv))][0], iv[int(0.95 * len(iv))][0]
busy = total = 0
cur_s = cur_e = None
for s, e in iv:
    s, e = max(s, lo), min(e, hi)
    if e <= s: continue
    total += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None: busy += cur_e - cur_s
print('window %.1f ms: busy %.1f %%, average kernels running %.2f (%d dispatches in the trace)' % ((hi - lo) / 1e6, 100.0 * busy / (hi - lo), total / (hi - lo), len(iv)))
