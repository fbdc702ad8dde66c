#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ap
mkdir -p $O
timeout 600 python scripts/copy_sites.py > $O/copy_sites.txt 2>&1; grep -E "^\s+[0-9]+ aten" $O/copy_sites.txt | head -32
