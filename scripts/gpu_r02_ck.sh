#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ck
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_models.py -q -k "gradient_checkpointing" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -n "^E  " $O/pytest.log | head -8
