#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02y3
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 1 --no-kernel-timer --eager --steps 3 --warmup 2"
cd /tmp
SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_PRIO0=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python $GRAFT_REPO_ROOT/bench.py $B > $O/tr.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r02y3/tr/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step only: take the last 1500 kernels
rows = rows[-1400:]
t0 = int(rows[0]['Start_Timestamp'])
qs = collections.Counter(r['Queue_Id'] for r in rows)
print('queues', qs)
main_q = qs.most_common(1)[0][0]
prev_end = None
gaps = []
for r in rows:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if r['Queue_Id'] != main_q:
        print(f"OTHER q={r['Queue_Id']} {s/1e3:10.1f} us .. {e/1e3:10.1f} us  {r['Kernel_Name'][:70]}")
        continue
    if prev_end is not None and s - prev_end > 100000:
        gaps.append((s - prev_end, prev_end, r['Kernel_Name'][:60], prev_name))
        print(f"GAP {(s-prev_end)/1e3:9.1f} us before {r['Kernel_Name'][:60]} at {s/1e3:10.1f} us (after {prev_name})")
    prev_end, prev_name = e, r['Kernel_Name'][:50]
print('total gap ms', sum(g[0] for g in gaps) / 1e6, 'span ms', (int(rows[-1]['End_Timestamp']) - t0) / 1e6)
PY
rm -rf $O/tr
