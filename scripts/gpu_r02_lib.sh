#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02lib
mkdir -p $O
KB_ITERS=8 timeout 200 python scripts/linear_bench.py 2>/dev/null | grep '^{' > $O/linear_vs_library.jsonl; cat $O/linear_vs_library.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['M'],d['K'],d['N'],'ours fwd',d['fwd_tf'],'dgrad',d['dgrad_tf'],'wgrad',d['wgrad_tf'],'| library fwd',d['lib_fwd_tf'],'wgrad',d['lib_wgrad_tf'])"
