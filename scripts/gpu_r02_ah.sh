#!/bin/bash
# r02: batched weight repack: tests + bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ah
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_train_loop.py tests/test_gpu_optim.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -n "^E " $O/pytest.log | head
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
for m in resnet50 vit_base_patch16; do for f in 1 0; do
  SAICV_PACK_BATCH=$f timeout 600 python bench.py --model $m $B > $O/bench_${m}_$f.log 2>&1; echo "$m batched repack=$f: $(grep '^{"metric' $O/bench_${m}_$f.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done
