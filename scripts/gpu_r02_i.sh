#!/bin/bash
# r02: config-driven DETR / full-SAM loop workloads, SAM encoder at the reference's per-GPU batch 20
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02i
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02i
B="--no-cpu-baseline --no-secondary --max-windows 2 --no-kernel-timer"
run() { name=$1; shift; timeout 900 python bench.py "$@" $B > $O/$name.log 2>&1; echo "$name rc=$? $(tail -1 $O/$name.log | cut -c1-420)"; python - <<PY
import torch
PY
}
run sam_b_full_b8 --model sam_b --batch 8 --steps 3 --warmup 2
run detr_config_b8 --model resnet50_detr_config --batch 8 --steps 4 --warmup 2
run sam_enc_b20 --model sam_b_encoder --batch 20 --steps 3 --warmup 2
run sam_b_full_b20 --model sam_b --batch 20 --steps 2 --warmup 1
run detr_b8 --model resnet50_detr --batch 8 --steps 4 --warmup 2
