#!/bin/bash
# HBM traffic of the default bench command: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) in eager mode, folded by
# scripts/make_pmc_summary.py into profiles/r03_pmc_hbm_traffic.json (copy it from gpurun_out/r03pmc/)
O=$GRAFT_REPO_ROOT/gpurun_out/r03pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O -o $c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-secondary --no-kernel-timer --max-windows 1 > $O/$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
python scripts/make_pmc_summary.py $O 3 $O/r03_pmc_hbm_traffic.json | head -60
