#!/bin/bash
# r02: per-layer ResNet-50 convolution times against the per-layer lower bound
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02m
mkdir -p $O
KB_ITERS=10 python scripts/kernel_bench.py 256 > $O/kernel_microbench.jsonl 2>$O/err.log
cat $O/kernel_microbench.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'conv' in d: print('%-24s x%d fwd %6.1f/%5.1f (%.2f) dgrad %6.1f/%5.1f (%.2f) wgrad %6.1f/%5.1f (%.2f)'%(d['conv'],d['count'],d['fwd_us'],d['fwd_bound_us'],d['fwd_frac_of_bound'],d['dgrad_us'],d.get('dgrad_bound_us',0),d.get('dgrad_frac_of_bound',0),d['wgrad_us'],d['wgrad_bound_us'],d['wgrad_frac_of_bound']))
    else: print(d)
"
tail -3 $O/err.log
