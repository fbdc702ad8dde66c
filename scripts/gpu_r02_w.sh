#!/bin/bash
# r02: DETR after LinearRowsFn: parity tests, bench, and host call sites of small ATen ops (cProfile callers)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02w
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_detr.py -x -q > $O/pytest_detr.log 2>&1; tail -2 $O/pytest_detr.log
timeout 600 python bench.py --model resnet50_detr_config --batch 8 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --max-windows 3 --no-kernel-timer > $O/bench_detr.log 2>&1; tail -1 $O/bench_detr.log | cut -c1-180
timeout 900 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--model', 'resnet50_detr_config', '--batch', '8', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--no-secondary', '--max-windows', '1', '--no-kernel-timer']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
for pat in ('built-in method torch.zeros', 'built-in method torch.empty}', 'built-in method torch.full', 'built-in method torch.ones', \"method 'to' of\", \"method 'float' of\", \"method 'contiguous' of\", \"method 'clone' of\", 'built-in method torch.cat', 'built-in method torch.stack', 'built-in method torch.tensor', \"method 'masked_fill\", \"method 'item' of\", \"method 'tolist' of\", 'built-in method apply'):
    st.print_callers(pat)
open('$O/detr_callers.txt', 'w').write(s.getvalue())
" > $O/prof.log 2>&1
grep -v "^$" $O/detr_callers.txt | grep -E "^\S|\s+[0-9]+" | awk 'length($0) < 260' | head -150
