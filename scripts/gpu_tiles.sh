#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for t in 0 1 2 3; do
  SAICV_NT_TILE=$t KB_SKIP_WGRAD=1 KB_CONV_ONLY=1 KB_ITERS=6 timeout 300 python scripts/kernel_bench.py 256 > gpurun_out/tiles_$t.jsonl 2> gpurun_out/tiles_$t.err
done
python - <<'PY'
import json
rows={}
for t in range(4):
    for line in open(f'gpurun_out/tiles_{t}.jsonl'):
        r=json.loads(line)
        if 'conv' in r:
            rows.setdefault(r['conv'],{})[t]=(r['fwd_us'],r['dgrad_us'])
for c,v in rows.items():
    f=[v[t][0] for t in range(4)]; d=[v[t][1] for t in range(4)]
    print(f"{c:28s} fwd {f} best {f.index(min(f))} | dgrad {d} best {d.index(min(d))}")
PY
