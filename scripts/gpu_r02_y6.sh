#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --eager --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
export SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_PRIO0=1 SAICV_DBG_NO_BCAST=1 SAICV_DBG_NORCCL=1
SAICV_DBG_NOJOINWAIT=1 timeout 600 python bench.py $B > $O/nojoinwait.log 2>&1; echo "bucket events, no wait on the compute stream: $(j $O/nojoinwait.log)"
SAICV_DBG_NOBUCKETEVENT=1 timeout 600 python bench.py $B > $O/joinonly.log 2>&1; echo "join only (no bucket events): $(j $O/joinonly.log)"
