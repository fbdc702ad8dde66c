#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r04.py -m gpu -q -s > $O/pytest_r04.log 2>&1; grep -E "vs float64|passed|failed" $O/pytest_r04.log | cut -c1-420 | head -20
timeout 1200 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-6000
bash scripts/gpu_pmc_r04.sh resnet50 vit_base_patch16 2>&1 | tail -60
