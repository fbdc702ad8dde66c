#!/bin/bash
# 3x3 staged-range convolution path: parity tests, then the ResNet-50 convolution micro-benchmark and the model with / without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03halo$1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers_b256.py -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for v in 0 1; do
  SAICV_NT_HALO=$v KB_SKIP_WGRAD=1 KB_CONV_ONLY=1 KB_ITERS=8 timeout 300 python scripts/kernel_bench.py 256 > $O/kb_$v.jsonl 2> $O/kb_$v.err
  python - <<PY
import json
for l in open('$O/kb_$v.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    if 'conv' in d and ' k3 s1' in d['conv']: print('halo=$v', d['conv'], 'fwd_us', d['fwd_us'], 'dgrad_us', d['dgrad_us'])
PY
  SAICV_NT_HALO=$v timeout 600 python bench.py --model resnet50 --no-secondary --no-cpu-baseline --max-windows 2 > $O/r50_$v.log 2>&1; tail -1 $O/r50_$v.log | cut -c1-160
done
