#!/bin/bash
# r04 call b: correctness of the touched epilogue paths, then the stagger / grouped-epilogue A/B on the fused benches
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_elemwise.py -m gpu -q -x > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
SAICV_NT_STAGGER2=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $O/pytest_kernels_stagger.log 2>&1; tail -2 $O/pytest_kernels_stagger.log
LF_ENVS="X=0;SAICV_NT_STAGGER2=2;SAICV_NT_STAGGER2=4;SAICV_NT_STAGGER2=8;SAICV_NT_STAGGER2=2,SAICV_NT_STAGGER2_PCT=60;SAICV_NT_STAGGER2=2,SAICV_NT_STAGGER2_PCT=140;SAICV_NT_KC8=3;SAICV_NT_KC8=3,SAICV_NT_STAGGER2=2" timeout 600 python scripts/linear_fused_bench.py > $O/linear_fused.jsonl 2> $O/linear_fused.err; grep vit_b_layer $O/linear_fused.jsonl; tail -2 $O/linear_fused.err
DG_ENVS="X=0;SAICV_NT_STAGGER2=2;SAICV_NT_STAGGER2=4;SAICV_NT_STAGGER2=2,SAICV_NT_STAGGER2_PCT=150" timeout 600 python scripts/dgrad_fused_bench.py > $O/dgrad_fused.jsonl 2> $O/dgrad_fused.err; grep model_weighted $O/dgrad_fused.jsonl; tail -2 $O/dgrad_fused.err
for v in "X=0" "SAICV_NT_STAGGER2=2" "SAICV_NT_STAGGER2=4"; do
  for m in resnet50 vit_base_patch16; do
    env $v timeout 600 python bench.py --model $m --no-secondary --no-cpu-baseline --max-windows 3 > $O/bench_${m}_$(echo $v | tr -c 'a-zA-Z0-9\n' '_').log 2>&1
    echo "$v $m: $(tail -1 $O/bench_${m}_$(echo $v | tr -c 'a-zA-Z0-9\n' '_').log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
  done
done
