#!/bin/bash
# SQ counters of the LayerNorm kernels inside the ViT-B step (eager, 2 steps)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --model vit_base_patch16 --steps 2 --warmup 1 --eager --no-cpu-baseline --no-secondary --no-kernel-timer --max-windows 1 > $O/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04w'
f=glob.glob(f'{O}/pmc/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'layernorm' not in k and 'row_scale' not in k and 'colsum' not in k and 'pack_weight' not in k: continue
    k=k[:60]+' g'+r.get('Grid_Size','')
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k, {c: round(x/n[(k,c)]/1e6,3) for c,x in v.items()})
P
