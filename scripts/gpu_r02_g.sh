#!/bin/bash
# r02: the new parity tests (batch-256 layers, real-dimension fixtures, f2 backbones, full gradients), then the whole suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02g
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02g
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|worst|trajectory|step graph\]|full gradients" $O/pytest_gpu.log | tail -40
