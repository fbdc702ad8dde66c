#!/bin/bash
# SAM / streaming attention parity on the GPU box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sam.py -x -q 2>&1 | tail -40 > gpurun_out/sam_tests.log
cat gpurun_out/sam_tests.log
