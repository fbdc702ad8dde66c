#!/bin/bash
# same-box A/B of the default bench line: the r02 tree (in _r02/, built in the container) against this tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03ab$1; mkdir -p $O
for rep in 1 2; do
  (cd _r02 && timeout 600 python bench.py --no-cpu-baseline --no-secondary --max-windows 5 > $O/r02_r50_$rep.log 2>&1; tail -1 $O/r02_r50_$rep.log | cut -c1-140)
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --max-windows 5 > $O/new_r50_$rep.log 2>&1; tail -1 $O/new_r50_$rep.log | cut -c1-140
  SAICV_NT_PERSIST=0 timeout 600 python bench.py --no-cpu-baseline --no-secondary --max-windows 5 > $O/new_np_r50_$rep.log 2>&1; tail -1 $O/new_np_r50_$rep.log | cut -c1-140
done
(cd _r02 && timeout 600 python bench.py --model vit_base_patch16 --no-cpu-baseline --no-secondary --max-windows 5 > $O/r02_vit.log 2>&1; tail -1 $O/r02_vit.log | cut -c1-140)
timeout 600 python bench.py --model vit_base_patch16 --no-cpu-baseline --no-secondary --max-windows 5 > $O/new_vit.log 2>&1; tail -1 $O/new_vit.log | cut -c1-140
SAICV_NT_PERSIST=0 timeout 600 python bench.py --model vit_base_patch16 --no-cpu-baseline --no-secondary --max-windows 5 > $O/new_np_vit.log 2>&1; tail -1 $O/new_np_vit.log | cut -c1-140
