#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --eager --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
SAICV_DBG_STREAM=1 timeout 600 python bench.py $B > $O/stream_plain.log 2>&1; echo "created compute stream, no DDP machinery: $(j $O/stream_plain.log)"
export SAICV_DDP_FORCE_SYNC=1
SAICV_DBG_STREAM=1 timeout 600 python bench.py $B > $O/stream_ddp.log 2>&1; echo "created compute stream, forced DDP sync (full native path): $(j $O/stream_ddp.log)"
SAICV_DBG_STREAM=1 SAICV_DBG_PRIO0=1 timeout 600 python bench.py $B > $O/stream_ddp_p0.log 2>&1; echo "same, normal-priority communication stream: $(j $O/stream_ddp_p0.log)"
