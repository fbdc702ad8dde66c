#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_loop.py -m gpu -q -x -s -k "detection_step_graph" > $O/pytest_detgraph.log 2>&1; grep -E "step graph|passed|failed|Error" $O/pytest_detgraph.log | cut -c1-300 | tail -8
timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -x -k "captured" > $O/pytest_ddp.log 2>&1; tail -4 $O/pytest_ddp.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_sam.py -m gpu -q -x > $O/pytest_sam.log 2>&1; tail -2 $O/pytest_sam.log | cut -c1-300
for m in resnet50_retinanet; do
  for e in "" "--eager"; do
    timeout 600 python bench.py --model $m $e --no-secondary --no-cpu-baseline --max-windows 3 > $O/bench_${m}_${e#--}.log 2>&1
    echo "$m $e: $(tail -1 $O/bench_${m}_${e#--}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["step_graph"], d["config"].get("host_enqueue_ms_per_step"))' 2>&1 | tail -1)"
  done
done
timeout 600 python bench.py --model sam_b_encoder --batch 20 --steps 5 --warmup 2 --max-windows 2 --no-secondary --no-cpu-baseline > $O/bench_sam_enc.log 2>&1; tail -1 $O/bench_sam_enc.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("sam_b_encoder", d["ms_per_step"], d["value"], d.get("kernel_breakdown_ms_per_step"))'
