#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --model vit_base_patch16 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_vit.log 2>&1
echo "rc=$?" >> gpurun_out/bench_vit.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_vit -o vit -- python $GRAFT_REPO_ROOT/bench.py --model vit_base_patch16 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/rocprof_vit.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/bench_vit.log | cut -c1-1200
