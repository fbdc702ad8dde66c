#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py -m gpu -q -x -k "stream_attention" > $O/pytest_attn.log 2>&1; tail -5 $O/pytest_attn.log | cut -c1-300
for v in 0 1; do
  echo "== SAICV_SA_FWD2=$v"; SAICV_SA_FWD2=$v timeout 600 python scripts/attn_bench.py 2>&1 | grep case | cut -c1-200
done
timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -x -k "captured" > $O/pytest_ddp.log 2>&1; tail -4 $O/pytest_ddp.log | cut -c1-400
