#!/bin/bash
# First GPU validation pass: build check, GPU parity tests, smoke, microbench, bench, rocprof.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx9" | head -4 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python scripts/kernel_bench.py 256 > gpurun_out/kbench.jsonl 2> gpurun_out/kbench.err
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head
tail -5 gpurun_out/pytest_gpu.log
tail -3 gpurun_out/smoke.log
tail -2 gpurun_out/bench.log
