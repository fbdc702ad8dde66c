#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["step_graph"])'; }
export SAICV_DDP_FORCE_SYNC=1
timeout 600 python bench.py $B > $O/graph_ddp.log 2>&1; echo "whole-step graph with the forced DDP sync captured: $(j $O/graph_ddp.log)"; grep -i "error\|fall\|warn" $O/graph_ddp.log | head -5
