#!/bin/bash
# r02: the driver's launch form (torch.distributed.run) at one rank, and the eager data-parallel path in a world of one
# (every bucket / event / join of the RCCL path, a mean over one rank) against the graph-replayed default
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02x
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 $B > $O/torchrun1.log 2>&1
echo "torchrun N=1: rc=$? $(tail -1 $O/torchrun1.log | cut -c1-130)"
for m in resnet50 vit_base_patch16; do
  timeout 600 python bench.py --model $m --eager $B > $O/eager_$m.log 2>&1; echo "eager $m: $(tail -1 $O/eager_$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("host_ms_per_step"))')"
  SAICV_DDP_FORCE_SYNC=1 timeout 600 python bench.py --model $m --eager $B > $O/ddp1_$m.log 2>&1; echo "eager + forced DDP sync $m: $(tail -1 $O/ddp1_$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("host_ms_per_step"))')"
done
