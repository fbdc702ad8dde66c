#!/bin/bash
# 128-byte K slices of the forward / data-gradient kernel: parity tests with the switch on, then isolated GEMMs and model benches
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03kc8$1; shift
mkdir -p $O
SAICV_NT_KC8=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers_b256.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in 0 1; do
  echo "== SAICV_NT_KC8=$v"
  SAICV_NT_KC8=$v timeout 300 python scripts/linear_bench.py > $O/lbench_$v.jsonl 2> $O/lbench_$v.err
  python - <<PY
import json
for l in open('$O/lbench_$v.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    print(d['M'], d['K'], d['N'], 'fwd_tf', d['fwd_tf'], 'dgrad_tf', d['dgrad_tf'], 'wgrad_tf', d['wgrad_tf'], 'lib_fwd', d['lib_fwd_tf'])
PY
  for m in "$@"; do
    SAICV_NT_KC8=$v timeout 600 python bench.py --model $m --no-secondary --no-cpu-baseline --max-windows 2 > $O/${m}_$v.log 2>&1; tail -1 $O/${m}_$v.log | cut -c1-160
  done
done
