#!/bin/bash
# HBM traffic of the default bench workloads: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) in eager mode per model, folded by
# scripts/make_pmc_summary.py into profiles/r04_pmc_hbm_traffic[_<model>].json (copy them from gpurun_out/r04pmc/)
#   gpu_pmc_r04.sh [model ...]      default: resnet50 vit_base_patch16
O=$GRAFT_REPO_ROOT/gpurun_out/r04pmc
mkdir -p $O
export TMPDIR=/tmp
MODELS="${@:-resnet50 vit_base_patch16}"
for m in $MODELS; do
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$m -o $c -- python $GRAFT_REPO_ROOT/bench.py --model $m --steps 2 --warmup 1 --eager --no-cpu-baseline --no-secondary --no-kernel-timer --max-windows 1 > $O/${m}_$c.log 2>&1; echo "pmc $m $c rc=$?"
  done
  cd $GRAFT_REPO_ROOT
  suffix=""; [ "$m" != "resnet50" ] && suffix="_$m"
  python scripts/make_pmc_summary.py $O/$m 3 $O/r04_pmc_hbm_traffic$suffix.json $m | head -40
done
