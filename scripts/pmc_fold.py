"""Fold rocprofv3 --pmc counter_collection CSVs of one or more passes into a JSON: per kernel family (short name + grid size class)
the per-launch averages and the sums, plus, when a kernel trace sits beside the counters, the launch durations.

    python scripts/pmc_fold.py <pass dir> [<pass dir> ...] [--by-grid]   > out.json
Families: the demangled kernel name cut at the first '<' (template arguments dropped) unless --by-grid also splits by grid size."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
by_grid = '--by-grid' in sys.argv


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'_ZN\d+_GLOBAL__N_1(\d+)', name)
    if m:                                   # mangled anonymous-namespace kernel: length-prefixed identifier
        n = int(m.group(1))
        rest = name[m.end():]
        name = rest[:n] + ' ' + rest[n:n + 60]
    return name[:90]


acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if by_grid:
                k += ' g' + r.get('Grid_Size', '?')
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[k][r['Counter_Name']] += 1
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if by_grid:
                k += ' g' + r.get('Grid_Size', '?')
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
out = {}
for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0.0]))):
    n = max(cnt[k].values())
    e = {'launches': n, 'sum_us': round(sum(dur.get(k, [])) / max(1, len(dirs)), 1)}
    for c in sorted(acc[k]):
        e[c] = round(acc[k][c] / max(cnt[k][c], 1), 1)
    if 'SQ_WAVE_CYCLES' in e and e['SQ_WAVE_CYCLES'] > 0:
        w = e['SQ_WAVE_CYCLES']
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM',
                  'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_INST_LDS'):
            if c in e:
                e[c + '_frac'] = round(e[c] / w, 3)
    if 'TCC_HIT_sum' in e and e['TCC_HIT_sum'] + e.get('TCC_MISS_sum', 0) > 0:
        e['l2_hit_rate'] = round(e['TCC_HIT_sum'] / (e['TCC_HIT_sum'] + e['TCC_MISS_sum']), 3)
    out[k] = e
json.dump(out, sys.stdout, indent=1)
