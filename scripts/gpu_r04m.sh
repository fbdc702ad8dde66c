#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py -m gpu -q -k "stream_attention" > $O/pytest_attn.log 2>&1; tail -2 $O/pytest_attn.log | cut -c1-300
for i in 1 2; do ATTN_CASES=plain_d64_n4096_b8,vit_b256_n197,sam_window_norel_b8 timeout 600 python scripts/attn_bench.py 2>&1 | grep case | cut -c1-200; done
ATTN_CASES=plain_d64_n4096_b8 bash scripts/gpu_r04k.sh | grep "^sa_"
python3 - <<'P'
import csv,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/r04k/pmc1/p1_kernel_trace.csv')):
    k=r['Kernel_Name']
    if 'sa_' in k: agg[k[k.index('sa_'):][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in agg.items(): print(k, len(v), round(sum(v)/len(v),1), round(min(v),1))
P
