#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04v; mkdir -p $O
for v in 1024 2048 4096 1024 2048; do
  SAICV_BN_BLOCKS=$v timeout 600 python bench.py --model resnet50 --no-secondary --no-cpu-baseline --no-sam --max-windows 3 > $O/bench_r50_$v.log 2>&1
  echo "SAICV_BN_BLOCKS=$v: $(tail -1 $O/bench_r50_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("kernel_breakdown_ms_per_step"))' 2>&1 | tail -1)"
done
