#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ar
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_layers_b256.py -x -q -k "atomic_rows" > $O/pytest.log 2>&1; tail -2 $O/pytest.log; grep -n "^E " $O/pytest.log | head -6
