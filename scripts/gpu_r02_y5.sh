#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --eager --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
export SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_PRIO0=1 SAICV_DBG_NO_BCAST=1
SAICV_DBG_FRESHEVENT=1 timeout 600 python bench.py $B > $O/fresh.log 2>&1; echo "fresh event per bucket: $(j $O/fresh.log)"
SAICV_DBG_LATE=1 timeout 600 python bench.py $B > $O/late.log 2>&1; echo "all buckets at the end of backward: $(j $O/late.log)"
SAICV_DBG_LATE=1 SAICV_DBG_NORCCL=1 timeout 600 python bench.py $B > $O/late_norccl.log 2>&1; echo "same, events only: $(j $O/late_norccl.log)"
