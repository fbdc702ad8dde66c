#!/bin/bash
# r06: the 256 x 256 tile on four wavefronts of 128 x 128 (library variants t22 = 16x16x32, t22m32 = 32x32x16) against the product
# library, ViT-B layer GEMMs (scripts/linear_fused_bench.py) and the ViT-B model, tile 0 forced and default picker
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06t22; mkdir -p $O
for rep in 1 2; do
for lib in - t22 t22m32; do
  SAICV_NT_TILE=0 timeout 600 python scripts/with_lib.py $lib scripts/linear_fused_bench.py > $O/lab_tile0_${lib}_$rep.jsonl 2> $O/lab_${lib}.err
  echo "lab tile0 lib [$lib] rep $rep: $(tail -1 $O/lab_tile0_${lib}_$rep.jsonl | cut -c1-400)"
done
done
timeout 600 python scripts/linear_fused_bench.py > $O/lab_default.jsonl 2>> $O/lab_-.err
echo "lab default picker: $(tail -1 $O/lab_default.jsonl | cut -c1-400)"
for rep in 1 2; do
for lib in - t22 t22m32; do
  SAICV_NT_TILE=0 timeout 600 python scripts/with_lib.py $lib bench.py --model vit_base_patch16 --no-secondary --no-cpu-baseline --no-sam --max-windows 2 > $O/vit_tile0_${lib}_$rep.log 2>&1
  echo "vit tile0 lib [$lib] rep $rep: $(tail -1 $O/vit_tile0_${lib}_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("kernel_breakdown_ms_per_step"), d.get("power"))' 2>&1 | cut -c1-400)"
done
done
timeout 600 python bench.py --model vit_base_patch16 --no-secondary --no-cpu-baseline --no-sam --max-windows 2 > $O/vit_default.log 2>&1
echo "vit default: $(tail -1 $O/vit_default.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("kernel_breakdown_ms_per_step"))' 2>&1 | cut -c1-400)"
