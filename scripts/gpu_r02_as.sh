#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02as
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_f2.py -x -q -k "vit_backbone" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -n "^E " $O/pytest.log | head -8; grep -n "worst gradient" $O/pytest.log | head
