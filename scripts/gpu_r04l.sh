#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py -m gpu -q -n 2 > $O/pytest_attn.log 2>&1; tail -3 $O/pytest_attn.log | cut -c1-300
for i in 1 2; do ATTN_CASES=sam_global_b8,plain_d64_n4096_b8 timeout 600 python scripts/attn_bench.py 2>&1 | grep case | cut -c1-200; done | tee $O/attn_bench.jsonl
