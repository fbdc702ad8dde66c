#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ao
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_models.py -q -k "world2 or native or arena_direct" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -n "^FAILED\|^E  " $O/pytest.log | head -8
