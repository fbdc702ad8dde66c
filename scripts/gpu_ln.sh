#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "layernorm or layer_norm or ln" 2>&1 | tail -2
for nb in 128 256 512 1024; do
SAICV_LN_BWD_BLOCKS=$nb python - <<PY 2>/dev/null
import torch, sys
sys.path.insert(0,'.')
from simpleaicv_pytorch_training_examples_amd import ops_tfm
from scripts.kernel_bench import timeit
M,C=50432,768
x=torch.randn(M,C,device='cuda').bfloat16(); dy=torch.randn(M,C,device='cuda').bfloat16(); ad=torch.randn(M,C,device='cuda').bfloat16()
w=torch.ones(C,device='cuda',requires_grad=True); b=torch.zeros(C,device='cuda',requires_grad=True)
y,mean,rstd=ops_tfm.ln_fwd(x,w,b,1e-6)
t=timeit(lambda: ops_tfm.ln_bwd(dy,x,w,b,mean,rstd,addend=ad))
print($nb, 'ln_bwd us', round(t*1e6,1), 'TB/s', round(4*M*C*2/t/1e12,2))
PY
done
