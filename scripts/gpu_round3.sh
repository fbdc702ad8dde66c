#!/bin/bash
# full GPU suite + the three bench lines (ResNet-50 headline with cpu_baseline, ViT-B/16, SAM encoder)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
timeout 600 python bench.py --model vit_base_patch16 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_vit.log 2>&1
timeout 600 python bench.py --model sam_b_encoder --batch 8 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_sam.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3
tail -3 gpurun_out/smoke.log
tail -1 gpurun_out/bench.log | cut -c1-900
tail -1 gpurun_out/bench_vit.log | cut -c1-400
tail -1 gpurun_out/bench_sam.log | cut -c1-400
