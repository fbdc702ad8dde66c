#!/bin/bash
# isolated linear GEMMs + one model bench per environment variant:  gpu_lb_env.sh <tag> <model> "<env 1>" "<env 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03lb$1; MODEL=$2; shift 2
mkdir -p $O
i=0
for v in "$@"; do
  i=$((i+1)); echo "== variant $i: $v"
  env $v timeout 300 python scripts/linear_bench.py > $O/lbench_$i.jsonl 2> $O/lbench_$i.err
  python - <<PY
import json
for l in open('$O/lbench_$i.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    print(d['M'], d['K'], d['N'], 'fwd_tf', d['fwd_tf'], 'dgrad_tf', d['dgrad_tf'], 'wgrad_tf', d['wgrad_tf'])
PY
  env $v timeout 600 python bench.py --model $MODEL --no-secondary --no-cpu-baseline --max-windows 2 > $O/${MODEL}_$i.log 2>&1; tail -1 $O/${MODEL}_$i.log | cut -c1-160
done
