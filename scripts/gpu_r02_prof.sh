#!/bin/bash
# r02 profiles: rocprofv3 kernel stats of the bench workloads (graph mode for R50 / ViT), copied to profiles/ by hand
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02prof
mkdir -p $O
B="--no-cpu-baseline --no-secondary --max-windows 1 --no-kernel-timer"
cd /tmp
prof() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python $GRAFT_REPO_ROOT/bench.py "$@" $B > $O/$name.log 2>&1; echo "$name rc=$? $(tail -1 $O/$name.log | cut -c1-160)"; }
prof r50 --model resnet50 --steps 5 --warmup 5
prof vit --model vit_base_patch16 --steps 5 --warmup 5
prof samfull --model sam_b --batch 20 --steps 2 --warmup 1
prof detrcfg --model resnet50_detr_config --batch 8 --steps 3 --warmup 2
cd $GRAFT_REPO_ROOT
rm -f $O/*/*kernel_trace.csv $O/*/*/*kernel_trace.csv
find $O -name "*stats*.csv" | head
