#!/bin/bash
# quick loop: GPU tests + microbench + bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python scripts/kernel_bench.py 256 > gpurun_out/kbench.jsonl 2> gpurun_out/kbench.err
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
tail -3 gpurun_out/kbench.jsonl
tail -1 gpurun_out/bench.log | cut -c1-900
