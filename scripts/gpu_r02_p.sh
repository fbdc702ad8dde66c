#!/bin/bash
# r02: rocprofv3 kernel stats of the ResNet-50 step with and without the fused BatchNorm-backward reduction (eager)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02p
mkdir -p $O
B="--no-cpu-baseline --no-secondary --max-windows 1 --no-kernel-timer --eager"
cd /tmp
prof() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python $GRAFT_REPO_ROOT/bench.py --model resnet50 --steps 5 --warmup 5 $B > $O/$name.log 2>&1; echo "$name rc=$? $(tail -1 $O/$name.log | cut -c1-120)"; }
SAICV_BN_FUSE=1 prof fuse1
SAICV_BN_FUSE=0 prof fuse0
cd $GRAFT_REPO_ROOT
rm -f $O/*/*kernel_trace.csv $O/*/*/*kernel_trace.csv
find $O -name "*kernel_stats.csv" | head
