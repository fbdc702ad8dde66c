#!/bin/bash
# HBM traffic of the bench itself: separate --pmc passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
mkdir -p gpurun_out/pmcb
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcb -o $c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/pmcb/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, json
out = {}
for c in ['FETCH_SIZE', 'WRITE_SIZE']:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f'gpurun_out/pmcb/{c}_counter_collection.csv')):
        if r['Counter_Name'] != c: continue
        n = r['Kernel_Name']
        key = 'igemm_nt' if 'igemm_nt' in n else 'igemm_tn' if 'igemm_tn' in n else 'bn_bwd_apply' if 'bn_bwd_apply' in n else 'bn_bwd_reduce' if 'bn_bwd_reduce' in n else 'bn_act_fwd' if 'bn_act_fwd' in n else None
        if key is None: continue
        agg[key][0] += 1
        agg[key][1] += float(r['Counter_Value'])
    out[c] = {k: {'launches': v[0], 'sum_counter_kb': v[1]} for k, v in agg.items()}
json.dump(out, open('gpurun_out/pmcb/traffic_summary.json', 'w'), indent=1)
print(json.dumps(out)[:1500])
PY
