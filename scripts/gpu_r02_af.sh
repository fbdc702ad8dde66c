#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02af
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bn_backward_fusions or gated_shortcut or dgrad_epilogue" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -n "^E " $O/pytest.log | head -12

