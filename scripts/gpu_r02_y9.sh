#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --eager --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
export SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_NO_BCAST=1
SAICV_COMM_MEMOPS=1 timeout 600 python bench.py $B > $O/memops.log 2>&1; echo "stream memory ops (write value / wait value), high-priority stream: $(j $O/memops.log)"; grep -i "error" $O/memops.log | head -3
SAICV_COMM_MEMOPS=1 SAICV_DBG_PRIO0=1 timeout 600 python bench.py $B > $O/memops_p0.log 2>&1; echo "same, normal priority: $(j $O/memops_p0.log)"
