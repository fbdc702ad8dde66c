#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B=${1:-8}
timeout 900 python bench.py --model sam_b_encoder --batch $B --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_sam.log 2>&1
echo "rc=$?" >> gpurun_out/bench_sam.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sam -o sam -- python $GRAFT_REPO_ROOT/bench.py --model sam_b_encoder --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/rocprof_sam.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/bench_sam.log | cut -c1-1500
python - <<'PY'
import csv,glob
fs=glob.glob('gpurun_out/prof_sam/**/*kernel_stats.csv',recursive=True)
if fs:
    rows=list(csv.DictReader(open(fs[0])))
    for r in rows[:28]:
        print(r['Name'][:90].ljust(90), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms", r['Percentage'])
PY
