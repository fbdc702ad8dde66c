#!/bin/bash
# r02: DETR loss vectorised over the decoder layers + host-side valid-row selection: parity tests, bench; JSON line last
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02aj
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_detr.py tests/test_gpu_train_loop.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log; grep -n "^E " $O/pytest.log | head -5
for m in "resnet50_detr_config --batch 8" "resnet50_detr --batch 8"; do
  n=$(echo $m | cut -d' ' -f1)
  timeout 600 python bench.py --model $m --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --max-windows 3 --no-kernel-timer > $O/bench_$n.log 2>&1; echo "$n: $(grep '^{"metric' $O/bench_$n.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
SAICV_DDP_FORCE_SYNC=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --max-windows 1 --no-kernel-timer --eager > $O/last_line.log 2>/dev/null; echo "last stdout line starts with: $(tail -1 $O/last_line.log | cut -c1-30)"
