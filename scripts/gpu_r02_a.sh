#!/bin/bash
# r02 first pass: full GPU test suite (new optimizer / trajectory / graph tests included), bench default line, eager A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02a
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.log 2>&1
echo "bench rc=$?"; tail -3 $O/bench_default.log | cut -c1-1800
timeout 600 python bench.py --eager --no-secondary --no-cpu-baseline --max-windows 3 > $O/bench_eager.log 2>&1
echo "eager rc=$?"; tail -1 $O/bench_eager.log | cut -c1-700
