#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_gpu_r04.py -m gpu -q -k "shortcut" > $O/pytest_join$i.log 2>&1; tail -4 $O/pytest_join$i.log | cut -c1-400
done
