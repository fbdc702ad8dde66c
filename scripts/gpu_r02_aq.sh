#!/bin/bash
# r02: maxpool_bwd with all window loads in flight; native-communicator test after the bootstrap refactor; bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02aq
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ddp.py tests/test_gpu_models.py -x -q -k "maxpool or native or resnet" > $O/pytest.log 2>&1; tail -2 $O/pytest.log; grep -n "^E " $O/pytest.log | head -5
KB_ITERS=10 timeout 300 python - > $O/maxpool.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from simpleaicv_pytorch_training_examples_amd import ops
x = torch.randn(256, 64, 112, 112, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
y = ops.max_pool2d(x, 3, 2, 1)
g = torch.randn_like(y)
for _ in range(3):
    x.grad = None; y.backward(g, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    x.grad = None; y.backward(g, retain_graph=True)
e1.record(); torch.cuda.synchronize()
print('maxpool_bwd (+ autograd glue) us per call', e0.elapsed_time(e1) * 100)
PY
tail -1 $O/maxpool.log
timeout 600 python bench.py --no-secondary --no-cpu-baseline --max-windows 3 --no-kernel-timer > $O/bench.log 2>&1; echo "r50: $(grep '^{"metric' $O/bench.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
