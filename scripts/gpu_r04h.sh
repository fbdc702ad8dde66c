#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r04.py -m gpu -q -x -k "shortcut" > $O/pytest_join.log 2>&1; tail -12 $O/pytest_join.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_ddp.py tests/test_gpu_decoders.py tests/test_gpu_retinanet.py tests/test_gpu_backbones.py -m gpu -q -k "resnet or captured or decode or retina or fcos or backbone" > $O/pytest_models.log 2>&1; tail -6 $O/pytest_models.log | cut -c1-300
for v in "SAICV_DS_JOIN_FUSE=0" "SAICV_DS_JOIN_FUSE=1" "SAICV_DS_JOIN_FUSE=0" "SAICV_DS_JOIN_FUSE=1"; do
  env $v timeout 600 python bench.py --model resnet50 --no-secondary --no-cpu-baseline --no-sam --max-windows 4 > $O/bench_r50_$v.log 2>&1
  echo "$v: $(tail -1 $O/bench_r50_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("kernel_breakdown_ms_per_step"))' 2>&1 | tail -1)"
done
