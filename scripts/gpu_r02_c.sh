#!/bin/bash
# r02 third pass: CU placement probe, GPU suite, persistent / anti-phase sweep of the NT kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02c
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/probes/cu_probe.hip -o /tmp/cu_probe > /dev/null 2>&1 && timeout 60 /tmp/cu_probe > $O/cu_probe.log 2>&1
head -30 $O/cu_probe.log
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|step graph\]" $O/pytest_gpu.log | tail -8
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
run() {  # name, env...
  name=$1; shift
  for m in resnet50 vit_base_patch16; do
    env "$@" timeout 600 python bench.py --model $m $B > $O/bench_${m}_${name}.log 2>&1
    echo "$name $m: $(tail -1 $O/bench_${m}_${name}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null || tail -2 $O/bench_${m}_${name}.log | cut -c1-300)"
  done
}
run off SAICV_NT_PERSIST=0
run dyn0 SAICV_NT_STAGGER_PCT=0
run st100 SAICV_NT_STAGGER_PCT=100
run st60 SAICV_NT_STAGGER_PCT=60
run st150 SAICV_NT_STAGGER_PCT=150
run st100min15 SAICV_NT_STAGGER_PCT=100 SAICV_NT_PERSIST_MIN_X10=15
run st100tile1 SAICV_NT_STAGGER_PCT=100 SAICV_NT_TILE=1
