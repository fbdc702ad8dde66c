#!/bin/bash
# A/B of the weight-gradient kernels on one box: SAICV_TN_DMA=0 (register-staged) vs 1 (LDS-DMA ring)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03tn$1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers_b256.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in 0 1; do
  SAICV_TN_DMA=$v timeout 300 python scripts/linear_bench.py > $O/lbench_$v.jsonl 2> $O/lbench_$v.err; python - <<PY
import json
for l in open('$O/lbench_$v.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    print('dma=$v', d['M'], d['K'], d['N'], 'wgrad_tf', d['wgrad_tf'], 'fwd_tf', d['fwd_tf'])
PY
  SAICV_TN_DMA=$v timeout 600 python bench.py --model vit_base_patch16 --no-secondary --no-cpu-baseline --max-windows 2 > $O/vit_$v.log 2>&1; tail -1 $O/vit_$v.log | cut -c1-200
  SAICV_TN_DMA=$v timeout 600 python bench.py --model resnet50 --no-secondary --no-cpu-baseline --max-windows 2 > $O/r50_$v.log 2>&1; tail -1 $O/r50_$v.log | cut -c1-200
done
