#!/bin/bash
# per-kernel SQ counters of the attention kernels (two passes; counters only, no other trace domain)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O
export ATTN_CASES=${ATTN_CASES:-sam_global_b8,plain_d64_n4096_b8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc1 -o p1 -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2 -o p2 -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py > $O/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name '*.csv' | head; 
python - <<'P'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04k'
for d in ('pmc1','pmc2'):
    f=glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True)
    if not f: print(d,'no csv'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']
        if 'sa_' not in k: continue
        k=k[k.index('sa_'):][:44]+' g'+r.get('Grid_Size','')
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        print(k, {c: round(x/n[(k,c)]/1e6,2) for c,x in v.items()})
P
