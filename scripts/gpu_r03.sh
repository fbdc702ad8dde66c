#!/bin/bash
# Round-3 GPU runner: one parametrised script instead of one file per experiment.
#   gpurun --timeout N -- 'bash scripts/gpu_r03.sh <tag> <steps...>'
# steps: tests[:<pytest args>]  smoke  sweep[:linear|conv]  kbench  lbench  bench[:<bench.py args>]  prof:<model>  pmc:<model>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/r03$TAG
mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case $name in
    tests)  timeout 1500 python -m pytest tests -m gpu -q -x ${arg:-} > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log ;;
    testsall) timeout 1800 python -m pytest tests -m gpu -q ${arg:-} > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    sweep)  timeout 900 python scripts/nt_sweep.py ${arg:-all} > $O/nt_sweep_${arg:-all}.jsonl 2> $O/nt_sweep.err; tail -3 $O/nt_sweep_${arg:-all}.jsonl | cut -c1-600; tail -3 $O/nt_sweep.err ;;
    kbench) KB_ITERS=10 timeout 600 python scripts/kernel_bench.py > $O/kernel_microbench.jsonl 2> $O/kbench.err; tail -4 $O/kernel_microbench.jsonl | cut -c1-400 ;;
    lbench) timeout 600 python scripts/linear_bench.py > $O/linear_bench.jsonl 2> $O/lbench.err; cat $O/linear_bench.jsonl | cut -c1-300 ;;
    bench)  timeout 900 python bench.py ${arg:-} > $O/bench_$(echo "${arg:-default}" | tr -c 'a-zA-Z0-9\n' '_').log 2>&1; tail -1 $O/bench_$(echo "${arg:-default}" | tr -c 'a-zA-Z0-9\n' '_').log | cut -c1-700 ;;
    prof)   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$arg -o $arg -- python $GRAFT_REPO_ROOT/bench.py --model $arg --no-secondary --no-cpu-baseline --max-windows 2 --steps 5 --warmup 5 > $O/prof_$arg.log 2>&1); f=$(find $O/prof_$arg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${arg}_rocprofv3_kernel_stats.csv && head -12 $f | cut -c1-200 ;;
    *) echo "unknown step $step" ;;
  esac
done
