#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sam.py -m gpu -q -k "stream_attention" > $O/pytest_attn.log 2>&1; tail -4 $O/pytest_attn.log | cut -c1-300
timeout 600 python scripts/attn_bench.py 2>&1 | grep case | cut -c1-200 | tee $O/attn_bench.jsonl
SAICV_SA_FWD2=2 ATTN_CASES=sam_window_b8 timeout 600 python scripts/attn_bench.py 2>&1 | grep case | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_detr.py tests/test_gpu_kernels.py -m gpu -q -k "attention or attn or mha or transformer" > $O/pytest_detr.log 2>&1; tail -3 $O/pytest_detr.log | cut -c1-300
ATTN_CASES=sam_global_b8,plain_d64_n4096_b8 bash scripts/gpu_r04k.sh | grep "^sa_"
python3 - <<'P'
import csv,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/r04k/pmc1/p1_kernel_trace.csv')):
    k=r['Kernel_Name']
    if 'sa_' in k: agg[k[k.index('sa_'):][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in agg.items(): print(k, len(v), round(sum(v)/len(v),1), round(min(v),1))
P
