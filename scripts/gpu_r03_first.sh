cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/r03a/pytest_kernels.log 2>&1; tail -5 gpurun_out/r03a/pytest_kernels.log
bash scripts/gpu_r03.sh a testsall sweep:linear sweep:conv
