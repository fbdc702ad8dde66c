"""Per-kernel sums of rocprofv3 --pmc counter_collection CSVs: python pmc_summarize.py <dir> [name filter]"""
import csv
import glob
import sys
from collections import defaultdict

d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if flt not in k:
            continue
        k = k.replace('(anonymous namespace)::', '')[:100] + ' grid=' + r.get('Grid_Size', '?')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f'    {c:32s} {acc[k][c] / max(cnt[k][c], 1):18.1f}  (avg of {cnt[k][c]})')
