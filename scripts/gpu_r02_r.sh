#!/bin/bash
# r02: general grouped fused epilogue: kernel tests + benches
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1; tail -2 $O/pytest_kernels.log
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
for m in resnet50 vit_base_patch16; do timeout 600 python bench.py --model $m $B > $O/bench_$m.log 2>&1; echo "$m: $(tail -1 $O/bench_$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; done
timeout 600 python bench.py --model sam_b --batch 20 --steps 3 --warmup 2 $B > $O/bench_sam.log 2>&1; echo "sam_b: $(tail -1 $O/bench_sam.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
