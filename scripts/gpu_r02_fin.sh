#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02fin
timeout 150 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -x > gpurun_out/r02fin/pytest.log 2>&1; tail -2 gpurun_out/r02fin/pytest.log
