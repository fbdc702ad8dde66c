#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_f2.py -m gpu -q -rf --tb=short -k "dinov3" -s > $O/pytest_dino.log 2>&1; grep -v "^W2026" $O/pytest_dino.log | tail -25 | cut -c1-300
