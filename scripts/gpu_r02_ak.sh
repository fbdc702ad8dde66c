#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ak
mkdir -p $O
timeout 600 python scripts/aten_sites.py sam_b 4 > $O/sam_sites.txt 2>&1; grep -E "^\s+[0-9]+  |aten ops seen|--- ops" $O/sam_sites.txt | head -80
