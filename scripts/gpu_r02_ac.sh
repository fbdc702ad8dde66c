#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02ac
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --eager --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
SAICV_DBG_TORCH_EVENTS=1 timeout 600 python bench.py $B > $O/torch_events.log 2>&1; echo "no DDP machinery, three torch.cuda.Event().record() per step: $(j $O/torch_events.log)"
SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_RECORDONLY=1 timeout 600 python bench.py $B > $O/recordonly.log 2>&1; echo "forced DDP sync, buckets = a bare hipEventRecord on the compute stream: $(j $O/recordonly.log)"
SAICV_DDP_FORCE_SYNC=1 SAICV_DBG_RECORDONLY=0 timeout 600 python bench.py $B > $O/nothing.log 2>&1; echo "forced DDP sync, buckets = nothing (join + broadcast stay): $(j $O/nothing.log)"
