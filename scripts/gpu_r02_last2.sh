#!/bin/bash
# r02, closing call: the whole GPU test suite with the final code
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02last2
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; grep -n "^FAILED" $O/pytest_gpu.log | head
