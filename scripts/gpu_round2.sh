#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python scripts/kernel_bench.py 256 > gpurun_out/kbench.jsonl 2> gpurun_out/kbench.err
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
timeout 300 python scripts/torch_prof.py 64 > gpurun_out/torch_prof.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof | head
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3
tail -3 gpurun_out/smoke.log
tail -2 gpurun_out/bench.log
