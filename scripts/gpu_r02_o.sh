#!/bin/bash
# r02: fused dgrad epilogue (gated shortcut + BN-backward sums): kernel test, model parity both ways, bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02o
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dgrad_epilogue or conv_bn_act" > $O/pytest_kernels.log 2>&1; tail -5 $O/pytest_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_f2.py -x -q > $O/pytest_models.log 2>&1; tail -5 $O/pytest_models.log
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
for f in 1 0; do
  SAICV_BN_FUSE=$f timeout 600 python bench.py --model resnet50 $B > $O/bench_r50_fuse$f.log 2>&1
  echo "fuse=$f: $(tail -1 $O/bench_r50_fuse$f.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("final_loss"))')"
done
