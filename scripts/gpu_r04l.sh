#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py tests/test_gpu_detr.py -m gpu -q -k "attention or attn or mha or transformer or fp32_matches or bf16_tracks" -n 2 > $O/pytest_attn.log 2>&1; tail -4 $O/pytest_attn.log | cut -c1-300
timeout 600 python scripts/attn_bench.py 2>&1 | grep case | cut -c1-200 | tee $O/attn_bench.jsonl
