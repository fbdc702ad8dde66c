#!/bin/bash
# r02: native RCCL communicator test + DDP tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02n
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ddp.py -x -q -k native > $O/pytest.log 2>&1
grep -n "assert\|Error\|error\|^E " $O/pytest.log | head -30
