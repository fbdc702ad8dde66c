#!/bin/bash
# r02: DDP tests incl. the DETR multi-use case, CE / train-loop tests after the EMA / invalid-label changes
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02u
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_train_loop.py tests/test_gpu_kernels.py -q -k "ddp or world2 or native or train or softmax_ce or step_graph or loop" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
grep -n "^E " $O/pytest.log | head -10
