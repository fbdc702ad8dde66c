#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --eager --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
export SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_PRIO0=1 SAICV_DBG_NO_BCAST=1
SAICV_DBG_NOEVENT=1 timeout 600 python bench.py $B > $O/noevent.log 2>&1; echo "no events (RCCL calls only): $(j $O/noevent.log)"
SAICV_DBG_NORCCL=1 timeout 600 python bench.py $B > $O/norccl.log 2>&1; echo "events only (no RCCL call): $(j $O/norccl.log)"
SAICV_DBG_NORCCL=1 SAICV_DBG_NOEVENT=1 timeout 600 python bench.py $B > $O/neither.log 2>&1; echo "neither: $(j $O/neither.log)"
