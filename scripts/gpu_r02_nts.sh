#!/bin/bash
# r02: non-temporal stores of the GEMM output tile (what the vendor library's kernels do): isolated GEMMs + ResNet-50 / ViT
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02nts
mkdir -p $O
for f in 0 1; do
  SAICV_NT_STORE=$f KB_ITERS=6 timeout 100 python scripts/linear_bench.py 2>/dev/null | grep '^{' | python -c "
import sys,json
print('nt_store=$f', ' '.join(str(json.loads(l)['fwd_tf'])+'/'+str(json.loads(l)['dgrad_tf']) for l in sys.stdin))"
done
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --steps 10 --warmup 5"
for f in 0 1; do for m in resnet50 vit_base_patch16; do
  SAICV_NT_STORE=$f timeout 200 python bench.py --model $m $B > $O/b_${m}_$f.log 2>&1; echo "nt_store=$f $m: $(grep '^{"metric' $O/b_${m}_$f.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done
