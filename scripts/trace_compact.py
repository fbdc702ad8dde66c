"""Compact a rocprofv3 kernel_trace.csv: one line per dispatch in start order -- short kernel name, duration (us), gap to the
previous dispatch's end (us), grid and workgroup size -- so that launches can be attributed to layers by their order."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prev_end = None
print('idx,name,dur_us,gap_us,grid,wg,vgpr,lds')
for i, r in enumerate(rows):
    n = r['Kernel_Name']
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = n.split('(')[0][:110].replace(',', ';')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = max(e, prev_end or e)
    print(f"{i},{n},{(e - s) / 1e3:.1f},{gap:.1f},{r.get('Grid_Size_X', r.get('Grid_Size', ''))},{r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))},"
          f"{r.get('VGPR_Count', '')},{r.get('LDS_Block_Size', '')}")
