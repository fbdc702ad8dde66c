#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py -m gpu -q -k "stream_attention" > $O/pytest_attn.log 2>&1; tail -8 $O/pytest_attn.log | cut -c1-300
SAICV_SA_FWD2=2 timeout 900 python -m pytest tests/test_gpu_sam.py -m gpu -q -k "stream_attention" > $O/pytest_attn_fwd2all.log 2>&1; tail -8 $O/pytest_attn_fwd2all.log | cut -c1-300
for v in 1 2; do
  echo "== SAICV_SA_FWD2=$v"; SAICV_SA_FWD2=$v timeout 600 python scripts/attn_bench.py 2>&1 | grep case | cut -c1-200 | tee $O/attn_bench_fwd2_$v.jsonl
done
timeout 900 python -m pytest tests/test_gpu_detr.py tests/test_gpu_kernels.py -m gpu -q -k "attention or attn or mha or transformer" > $O/pytest_detr.log 2>&1; tail -5 $O/pytest_detr.log | cut -c1-300
