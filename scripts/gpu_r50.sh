#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1
echo "rc=$?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r50 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/bench.log | cut -c1-1500
