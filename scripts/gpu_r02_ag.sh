#!/bin/bash
# r02: stem on the space-to-depth image: kernel test, model parity, bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ag
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "stem_on_the_space or conv_bn_act" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -n "^E " $O/pytest.log | head
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_detr.py tests/test_gpu_f2.py -x -q > $O/pytest_models.log 2>&1; tail -2 $O/pytest_models.log; grep -n "^E " $O/pytest_models.log | head -5
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
for f in 1 0; do
  SAICV_STEM_S2D=$f timeout 600 python bench.py $B > $O/bench_s2d$f.log 2>&1; echo "stem s2d=$f: $(grep '^{"metric' $O/bench_s2d$f.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
