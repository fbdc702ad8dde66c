#!/bin/bash
# r02 second pass: full GPU suite (no -x), weight-gradient side stream A/B on ResNet-50 and ViT-B (graph and eager)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02b
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|trajectory" $O/pytest_gpu.log | tail -15
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
for side in 1 0; do
  for m in resnet50 vit_base_patch16; do
    SAICV_WGRAD_SIDE=$side timeout 600 python bench.py --model $m $B > $O/bench_${m}_side${side}.log 2>&1
    echo "side=$side $m graph: $(tail -1 $O/bench_${m}_side${side}.log | cut -c1-130)"
  done
done
SAICV_WGRAD_SIDE=1 timeout 600 python bench.py --eager $B > $O/bench_r50_side1_eager.log 2>&1
echo "side=1 r50 eager: $(tail -1 $O/bench_r50_side1_eager.log | cut -c1-130)"
