#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --eager --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_NO_BCAST=1 timeout 600 python bench.py $B > $O/nobcast.log 2>&1; echo "no per-forward broadcast: $(j $O/nobcast.log)"
SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_PRIO0=1 timeout 600 python bench.py $B > $O/prio0.log 2>&1; echo "normal-priority communication stream: $(j $O/prio0.log)"
SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_PRIO0=1 SAICV_DBG_NO_BCAST=1 timeout 600 python bench.py $B > $O/both.log 2>&1; echo "both: $(j $O/both.log)"
