#!/bin/bash
# r02: grouped tile order / setprio on the existing NT kernel (isolated GEMMs and the model benches)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02l
mkdir -p $O
lb() { name=$1; shift; env "$@" KB_ITERS=5 python scripts/linear_bench.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', d['K'], d['N'], 'fwd', d['fwd_tf'], 'dgrad', d['dgrad_tf'])"; }
lb base X=1
lb gm4 SAICV_NT_GM=4
lb gm8 SAICV_NT_GM=8
lb prio SAICV_NT_PRIO=1
lb gm8prio SAICV_NT_GM=8 SAICV_NT_PRIO=1
lb t0gm8 SAICV_NT_TILE=0 SAICV_NT_GM=8
lb t0 SAICV_NT_TILE=0
B="--no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer"
run() { name=$1; shift; for m in resnet50 vit_base_patch16; do env "$@" timeout 600 python bench.py --model $m $B > $O/bench_${m}_${name}.log 2>&1; echo "$name $m: $(tail -1 $O/bench_${m}_${name}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; done; }
run base X=1
run gm8 SAICV_NT_GM=8
run prio SAICV_NT_PRIO=1
run base2 X=1
