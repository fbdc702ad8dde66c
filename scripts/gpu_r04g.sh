#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r04.py -m gpu -q -x -k "stem" > $O/pytest_stem.log 2>&1; tail -12 $O/pytest_stem.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "resnet" > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log | cut -c1-300
for v in "SAICV_STEM_POOL_FUSE=0" "SAICV_STEM_POOL_FUSE=1"; do
  env $v timeout 600 python bench.py --model resnet50 --no-secondary --no-cpu-baseline --max-windows 4 > $O/bench_r50_$v.log 2>&1
  echo "$v: $(tail -1 $O/bench_r50_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("kernel_breakdown_ms_per_step"))' 2>&1 | tail -1)"
done
for v in "SAICV_NT_KC8=2"; do
  env $v timeout 600 python bench.py --model vit_base_patch16 --no-secondary --no-cpu-baseline --no-kernel-timer --max-windows 3 > $O/bench_vit_$v.log 2>&1
  echo "$v vit: $(tail -1 $O/bench_vit_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
