#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_input.py -m gpu -q -rf --tb=short > $O/pytest_input.log 2>&1; tail -15 $O/pytest_input.log | cut -c1-300
