#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ab
mkdir -p $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^{\|Warn\|warn" $1 | head -22; }
timeout 600 python scripts/ddp_timeline.py > $O/timeline_base.txt 2>&1; echo "== baseline (no DDP machinery)"; f $O/timeline_base.txt
SAICV_DDP_FORCE_SYNC=1 SAICV_COMM_MODE=events timeout 600 python scripts/ddp_timeline.py > $O/timeline_events.txt 2>&1; echo "== forced DDP sync, event form"; f $O/timeline_events.txt
