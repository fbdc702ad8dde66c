#!/bin/bash
# rocprofv3 kernel stats of the ResNet-50 and ViT-B/16 bench steps (quick look; the evidence run is gpu_round5.sh)
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e -o r50 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/rocprof_e.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e -o vit -- python $GRAFT_REPO_ROOT/bench.py --model vit_base_patch16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/rocprof_ev.log 2>&1
rm -f $O/prof_e/*kernel_trace.csv
