#!/bin/bash
# r02: the data-parallel path with the helper-thread communicator (host-side waits): tests, then eager / graph at world 1
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02aa
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ddp.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["step_graph"])'; }
timeout 600 python bench.py $B --eager > $O/eager.log 2>&1; echo "eager, no DDP machinery: $(j $O/eager.log)"
export SAICV_DDP_FORCE_SYNC=1
timeout 600 python bench.py $B --eager > $O/eager_ddp.log 2>&1; echo "eager, forced DDP sync, helper thread: $(j $O/eager_ddp.log)"
SAICV_COMM_MODE=events timeout 600 python bench.py $B --eager > $O/eager_ddp_events.log 2>&1; echo "eager, forced DDP sync, event form: $(j $O/eager_ddp_events.log)"
timeout 600 python bench.py $B > $O/graph_ddp.log 2>&1; echo "graph, forced DDP sync: $(j $O/graph_ddp.log)"
timeout 600 python bench.py --model vit_base_patch16 $B --eager > $O/eager_ddp_vit.log 2>&1; echo "ViT eager, forced DDP sync, helper thread: $(j $O/eager_ddp_vit.log)"
