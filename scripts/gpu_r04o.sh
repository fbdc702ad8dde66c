#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_r04.py tests/test_gpu_elemwise.py tests/test_gpu_retinanet.py tests/test_gpu_retinaloss.py tests/test_gpu_fcosloss.py tests/test_gpu_optim.py tests/test_gpu_input.py tests/test_gpu_dwconv.py -m gpu -q -rf --tb=line --timeout 300 -n 3 --dist loadfile > $O/pytest_c.log 2>&1; tail -12 $O/pytest_c.log | cut -c1-300
