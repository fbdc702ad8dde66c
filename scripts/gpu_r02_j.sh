#!/bin/bash
# r02: SAM prompt-mask convolutions on the HIP GEMM path: parity, full-SAM bench, profile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py tests/test_gpu_f2.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|FAILED|^E  " $O/pytest.log | cut -c1-250 | tail
B="--no-cpu-baseline --no-secondary --max-windows 2 --no-kernel-timer"
timeout 900 python bench.py --model sam_b --batch 20 --steps 3 --warmup 2 $B > $O/sam_b_full_b20.log 2>&1; tail -1 $O/sam_b_full_b20.log | cut -c1-200
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/samfull -o samfull -- python $GRAFT_REPO_ROOT/bench.py --model sam_b --batch 20 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-windows 1 --no-kernel-timer > $O/samfull.log 2>&1
cd $GRAFT_REPO_ROOT; rm -f $O/samfull/*kernel_trace.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/samfull/samfull_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:14]: print(r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2))
PY
