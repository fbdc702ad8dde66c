#!/bin/bash
# r02 sixth pass: DMA issue interleaved with the MFMAs: parity, KC = 4 / 8, forced geometries, GEMM microbench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02f
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02f
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -8
for kc in 4 8; do for t in 0 1; do
  SAICV_NT_KC=$kc SAICV_NT_TILE=$t KB_ITERS=5 python scripts/linear_bench.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('kc$kc t$t', d['K'], d['N'], 'fwd', d['fwd_tf'], 'dgrad', d['dgrad_tf'], 'wgrad', d['wgrad_tf'])"
done; done
B="--no-secondary --no-cpu-baseline --max-windows 3"
run() {  # name, env...
  name=$1; shift
  for m in resnet50 vit_base_patch16; do
    env "$@" timeout 600 python bench.py --model $m $B > $O/bench_${m}_${name}.log 2>&1
    echo "$name $m: $(tail -1 $O/bench_${m}_${name}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("achieved"), d.get("kernel_breakdown_ms_per_step"))' 2>/dev/null || tail -2 $O/bench_${m}_${name}.log | cut -c1-300)"
  done
}
run kc4 SAICV_NT_KC=4
run kc8 SAICV_NT_KC=8
run kc8_t0 SAICV_NT_KC=8 SAICV_NT_TILE=0
run kc8_t1 SAICV_NT_KC=8 SAICV_NT_TILE=1
run kc8_t2 SAICV_NT_KC=8 SAICV_NT_TILE=2
run kc8_t3 SAICV_NT_KC=8 SAICV_NT_TILE=3
