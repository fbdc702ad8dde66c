#!/bin/bash
# r02: fused dgrad epilogue, grouped fast path: kernel tests, ViT sublayer tests, bench A/B, profile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02q
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
for f in 1 0; do
  SAICV_BN_FUSE=$f timeout 600 python bench.py --model resnet50 $B > $O/bench_r50_fuse$f.log 2>&1
  echo "fuse=$f: $(tail -1 $O/bench_r50_fuse$f.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
timeout 600 python bench.py --model vit_base_patch16 $B > $O/bench_vit.log 2>&1; echo "vit: $(tail -1 $O/bench_vit.log | cut -c1-100)"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fuse1 -o fuse1 -- python $GRAFT_REPO_ROOT/bench.py --model resnet50 --steps 5 --warmup 5 --no-cpu-baseline --no-secondary --max-windows 1 --no-kernel-timer --eager > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $O/*/*kernel_trace.csv
python scripts/prof_categories.py $O/fuse1/fuse1_kernel_stats.csv 10
