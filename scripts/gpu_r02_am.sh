#!/bin/bash
# r02: BatchNorm statistics as a few atomically accumulated rows, finalised inside the consuming kernels: tests + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02am
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_train_loop.py tests/test_gpu_f2.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -n "^E " $O/pytest.log | head -8
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
for f in 1 0; do
  SAICV_BN_INLINE=$f timeout 600 python bench.py $B > $O/bench_inline$f.log 2>&1; echo "inline=$f: $(grep '^{"metric' $O/bench_inline$f.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("host_enqueue_ms_per_step"))')"
done
