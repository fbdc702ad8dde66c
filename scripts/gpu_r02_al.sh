#!/bin/bash
# r02: multi-use parameters (SAM decoder) accumulate straight into the arena: SAM parity + DDP tests, SAM bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02al
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sam.py tests/test_gpu_ddp.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log; grep -n "^E " $O/pytest.log | head -5
timeout 600 python bench.py --model sam_b --batch 20 --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --max-windows 3 --no-kernel-timer > $O/bench_sam.log 2>&1; echo "sam_b b20: $(grep '^{"metric' $O/bench_sam.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
