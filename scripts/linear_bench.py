"""ViT-B linear-layer GEMM microbench (fwd / dgrad / wgrad) on one MI355X, with the vendor library (torch.matmul ->
hipBLASLt / rocBLAS, bf16) on the same shapes as the yardstick (not on the product path)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleaicv_pytorch_training_examples_amd import _lib
from simpleaicv_pytorch_training_examples_amd._lib import check, lib, ptr
from kernel_bench import timeit
L = lib(); st = _lib.stream()
M0 = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
for (M, K, N) in [(M0, 768, 2304), (M0, 768, 768), (M0, 768, 3072), (M0, 3072, 768), (M0, 4096, 4096), (4096, 4096, 4096), (8192, 8192, 8192)]:
    x = torch.randn(M, K, device='cuda').bfloat16(); wf = (torch.randn(N, K, device='cuda') * 0.03).bfloat16()
    wd = (torch.randn(K, N, device='cuda') * 0.03).bfloat16(); y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(M, N, device='cuda').bfloat16(); dx = torch.empty_like(x); dw = torch.zeros(N, K, device='cuda'); db = torch.zeros(N, device='cuda')
    fl = 2.0 * M * K * N
    tf = timeit(lambda: check(L.saicv_linear_fwd(0, ptr(x), ptr(wf), 0, ptr(y), M, K, N, 0, 0, 0, 1, st)))
    td = timeit(lambda: check(L.saicv_linear_dgrad(0, ptr(dy), ptr(wd), ptr(dx), M, K, N, 0, st)))
    tw = timeit(lambda: check(L.saicv_linear_wgrad(0, ptr(dy), ptr(x), ptr(dw), ptr(db), M, K, N, st)))
    import torch.nn.functional as F
    tl = timeit(lambda: F.linear(x, wf))                       # y = x W^T, the same NT product
    tlw = timeit(lambda: torch.matmul(dy.t(), x))              # weight-gradient product (bf16 output in the library)
    print(json.dumps({'M': M, 'K': K, 'N': N, 'lib_fwd_tf': round(fl/tl/1e12,1), 'lib_wgrad_tf': round(fl/tlw/1e12,1), 'fwd_us': round(tf*1e6,1), 'fwd_tf': round(fl/tf/1e12,1), 'dgrad_us': round(td*1e6,1), 'dgrad_tf': round(fl/td/1e12,1), 'wgrad_us': round(tw*1e6,1), 'wgrad_tf': round(fl/tw/1e12,1)}), flush=True)
