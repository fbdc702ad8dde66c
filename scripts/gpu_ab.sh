#!/bin/bash
# A/B of an environment switch on the three bench workloads:  bash scripts/gpu_ab.sh VAR
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1
for m in "resnet50 --batch 256 --steps 15 --warmup 4" "vit_base_patch16 --batch 256 --steps 8 --warmup 3" "sam_b_encoder --batch 8 --steps 4 --warmup 2"; do
  a=$(env $V=1 timeout 300 python bench.py --model $m --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  b=$(timeout 300 python bench.py --model $m --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "$m : $V=1 -> $a   default -> $b"
done
