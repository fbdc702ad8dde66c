#!/bin/bash
# end-of-round check: the GPU test suite, smoke(), then the r01e evidence set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r01e
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2 > gpurun_out/r01e/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r01e/smoke.log
cat gpurun_out/r01e/pytest_gpu.log gpurun_out/r01e/smoke.log
bash scripts/gpu_round5.sh
