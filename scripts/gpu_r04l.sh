#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py tests/test_gpu_detr.py tests/test_gpu_kernels.py -m gpu -q -n 3 > $O/pytest_attn.log 2>&1; tail -3 $O/pytest_attn.log | cut -c1-300
for i in 1 2; do timeout 600 python scripts/attn_bench.py 2>&1 | grep case | cut -c1-200; done | tee $O/attn_bench.jsonl
