#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02v
mkdir -p $O
timeout 900 python scripts/detr_op_profile.py resnet50_detr_config 8 > $O/detr_ops.txt 2>&1; tail -75 $O/detr_ops.txt | cut -c1-230
