#!/bin/bash
# r02: SAM tail kernels, re-gated gradient tests, f2 backbones
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02h
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02h
timeout 1200 python -m pytest tests/test_gpu_sam.py tests/test_gpu_f2.py tests/test_gpu_detr.py tests/test_gpu_models.py tests/test_gpu_ddp.py -m gpu -q -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|full gradients|^E  " $O/pytest_gpu.log | cut -c1-300 | tail -30
