#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|FAILED|^E  " $O/pytest.log | cut -c1-250 | tail -5
B="--no-cpu-baseline --no-secondary --max-windows 2 --no-kernel-timer"
timeout 900 python bench.py --model sam_b --batch 20 --steps 3 --warmup 2 $B > $O/sam_b_full_b20.log 2>&1; tail -1 $O/sam_b_full_b20.log | cut -c1-200
timeout 900 python bench.py --model sam_b --batch 8 --steps 3 --warmup 2 $B > $O/sam_b_full_b8.log 2>&1; tail -1 $O/sam_b_full_b8.log | cut -c1-200
