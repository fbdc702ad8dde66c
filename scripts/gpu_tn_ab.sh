#!/bin/bash
# weight-gradient kernels on one box: tests, then linear / conv micro-benchmarks and the two model benches per variant
#   usage: gpu_tn_ab.sh <tag> "<env assignments of variant 1>" "<env assignments of variant 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03tn$1; shift
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers_b256.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== variant $i: $v"
  env $v timeout 300 python scripts/linear_bench.py > $O/lbench_$i.jsonl 2> $O/lbench_$i.err
  python - <<PY
import json
for l in open('$O/lbench_$i.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    print(d['M'], d['K'], d['N'], 'wgrad_tf', d['wgrad_tf'], 'fwd_tf', d['fwd_tf'], 'dgrad_tf', d['dgrad_tf'])
PY
  env $v timeout 600 python bench.py --model vit_base_patch16 --no-secondary --no-cpu-baseline --max-windows 2 > $O/vit_$i.log 2>&1; tail -1 $O/vit_$i.log | cut -c1-200
  env $v timeout 600 python bench.py --model resnet50 --no-secondary --no-cpu-baseline --max-windows 2 > $O/r50_$i.log 2>&1; tail -1 $O/r50_$i.log | cut -c1-200
done
