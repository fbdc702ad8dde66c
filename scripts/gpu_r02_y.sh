#!/bin/bash
# r02: where the forced data-parallel path loses 13 ms per step at world 1
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
B="--no-secondary --no-cpu-baseline --max-windows 2 --no-kernel-timer --eager --steps 10 --warmup 3"
j() { grep '^{"metric' $1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
SAICV_DDP_FORCE_SYNC=1 SAICV_NATIVE_COMM=0 timeout 600 python bench.py $B > $O/hooks_only.log 2>&1; echo "hooks only (no communicator): $(j $O/hooks_only.log)"
SAICV_DDP_FORCE_SYNC=1 timeout 600 python bench.py $B > $O/native.log 2>&1; echo "native communicator: $(j $O/native.log)"
SAICV_DDP_FORCE_SYNC=1 timeout 900 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py'] + '$B'.split()
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
open('$O/cprofile_tot.txt', 'w').write(s.getvalue())
" > $O/prof.log 2>&1
head -45 $O/cprofile_tot.txt | cut -c1-180
