#!/bin/bash
# the secondary workloads' bench lines at the end of round 4 (one box for the table)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04s; mkdir -p $O
for m in resnet50_detr_config resnet50_detr sam_b resnet50_retinanet resnet50_fcos; do
  timeout 400 python bench.py --model $m --no-secondary --no-cpu-baseline --no-kernel-timer --max-windows 3 > $O/bench_$m.log 2>&1
  echo "$m: $(tail -1 $O/bench_$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("step_graph"), d.get("model_mfma_frac"))' 2>&1 | tail -1)"
done
