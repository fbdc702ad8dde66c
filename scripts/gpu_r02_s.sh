#!/bin/bash
# r02: where the DETR step's host time goes (cProfile of the bench worker), and the CIFAR config[0] epoch through the entry script
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02s
mkdir -p $O
timeout 900 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--model', 'resnet50_detr_config', '--batch', '8', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-secondary', '--max-windows', '1', '--no-kernel-timer']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(70)
open('$O/detr_cprofile_cum.txt', 'w').write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
open('$O/detr_cprofile_tot.txt', 'w').write(s.getvalue())
" > $O/detr_prof.log 2>&1
tail -2 $O/detr_prof.log | cut -c1-200
