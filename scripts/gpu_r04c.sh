#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r04.py -m gpu -q -s > $O/pytest_r04.log 2>&1; grep -E "vs float64|passed|failed|Error|error|assert" $O/pytest_r04.log | cut -c1-400 | head -60
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_elemwise.py -m gpu -q -x > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
for m in vit_base_patch16; do
  timeout 600 python bench.py --model $m --no-secondary --no-cpu-baseline --max-windows 3 > $O/bench_$m.log 2>&1
  echo "$m: $(tail -1 $O/bench_$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
