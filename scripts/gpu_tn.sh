#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad or linear or conv" 2>&1 | tail -4
echo "--- 128x128 tiles"; SAICV_TN_BIG=0 timeout 300 python scripts/linear_bench.py 2>&1 | cut -c1-200
echo "--- 256x256 tiles"; SAICV_TN_BIG=1 timeout 300 python scripts/linear_bench.py 2>&1 | cut -c1-200
echo "--- conv wgrad 128"; SAICV_TN_BIG=0 KB_CONV_ONLY=1 KB_ITERS=6 timeout 300 python scripts/kernel_bench.py 256 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'conv' in r: print(r['conv'], r['wgrad_us'], r['wgrad_tflops'])
"
echo "--- conv wgrad 256"; SAICV_TN_BIG=1 KB_CONV_ONLY=1 KB_ITERS=6 timeout 300 python scripts/kernel_bench.py 256 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'conv' in r: print(r['conv'], r['wgrad_us'], r['wgrad_tflops'])
"
