"""Average PMC counter values per dispatch of the kernels whose name contains a substring:
python tools/pmc_summary.py <counter_collection.csv> <substring>"""
import collections
import csv
import sys

agg, n = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value'])
        n[r['Counter_Name']] += 1
for k in sorted(agg):
    print('%-28s %16.0f  (%d dispatches)' % (k, agg[k] / n[k], n[k]))
