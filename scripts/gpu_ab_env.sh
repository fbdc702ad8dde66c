#!/bin/bash
# A/B of environment switches on one box:  gpu_ab_env.sh <tag> <pytest -k expression or -> <model> "<env variant 1>" "<env variant 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03ab$1; K=$2; MODEL=$3; shift 3
mkdir -p $O
if [ "$K" != "-" ]; then timeout 900 python -m pytest tests -m gpu -q -x -k "$K" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; fi
i=0
for v in "$@"; do
  i=$((i+1))
  for rep in 1 2; do
    env $v timeout 600 python bench.py --model $MODEL --no-secondary --no-cpu-baseline --max-windows 2 > $O/${MODEL}_${i}_$rep.log 2>&1
    echo "variant $i ($v) rep $rep: $(tail -1 $O/${MODEL}_${i}_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
  done
done
