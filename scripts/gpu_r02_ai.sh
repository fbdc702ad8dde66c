#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ai
mkdir -p $O
timeout 600 python scripts/detr_op_profile.py resnet50 256 > $O/r50_ops.txt 2>&1; grep -E "^\s+[0-9]+ " $O/r50_ops.txt | head -30
