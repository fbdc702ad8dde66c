#!/bin/bash
# lean GELU epilogue: parity tests, then same-box A/B on the ViT-B and SAM-B encoder benches
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_layers_b256.py -m gpu -q -n 3 -k "linear or gelu or vit or mlp" > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
for v in 0 1 0 1; do
  SAICV_GELU_EPI=$v timeout 600 python bench.py --model vit_base_patch16 --no-secondary --no-cpu-baseline --no-kernel-timer --max-windows 4 > $O/bench_vit_$v.log 2>&1
  echo "SAICV_GELU_EPI=$v vit: $(tail -1 $O/bench_vit_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
