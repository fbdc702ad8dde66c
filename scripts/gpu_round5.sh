#!/bin/bash
# r01e evidence: bench lines (ResNet-50 headline with cpu_baseline, ViT-B/16, SAM encoder, DETR-R50), rocprofv3 kernel stats of
# each, and the two PMC passes (FETCH_SIZE / WRITE_SIZE) over the headline bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r01e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r01e
timeout 900 python bench.py > $O/bench_resnet50.log 2>&1
timeout 600 python bench.py --model vit_base_patch16 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_vit.log 2>&1
timeout 600 python bench.py --model sam_b_encoder --batch 8 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_sam.log 2>&1
timeout 600 python bench.py --model resnet50_detr --batch 8 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_detr.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r50 -o r50 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/rocprof_r50.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vit -o vit -- python $GRAFT_REPO_ROOT/bench.py --model vit_base_patch16 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/rocprof_vit.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sam -o sam -- python $GRAFT_REPO_ROOT/bench.py --model sam_b_encoder --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > $O/rocprof_sam.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc -o $c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > $O/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
rm -f $O/prof_*/*kernel_trace.csv      # traces are large; the stats tables are what is kept
for m in resnet50 vit sam detr; do tail -1 $O/bench_$m.log | cut -c1-260; done
ls $O $O/pmc | head -30
