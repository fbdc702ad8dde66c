"""Depthwise convolution kernels: time and effective HBM rate at the depthwise layers of VAN-B2 / ConvFormer-S18 (batch 128, bf16).
    python scripts/dwconv_bench.py  ->  JSON lines (algorithmic bytes = read x / dy + write y / dx, 2 B per element)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpleaicv_pytorch_training_examples_amd import ops  # noqa: E402

# (name, N, C, H, k, stride, pad, dilation)
SHAPES = [('van stage1 3x3', 128, 64, 56, 3, 1, 1, 1), ('van stage1 lka 5x5', 128, 64, 56, 5, 1, 2, 1), ('van stage1 lka 7x7 d3', 128, 64, 56, 7, 1, 9, 3),
          ('van stage3 lka 7x7 d3', 128, 320, 14, 7, 1, 9, 3), ('convformer stage1 7x7', 128, 128, 56, 7, 1, 3, 1),
          ('convformer stage3 7x7', 128, 640, 14, 7, 1, 3, 1)]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for name, n, c, h, k, s, p, d in SHAPES:
    x = torch.randn(n, c, h, h, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(c, 1, k, k, device='cuda') / k).requires_grad_(True)
    y = ops.depthwise_conv2d(x, w, None, s, p, d)
    dy = torch.randn_like(y)
    t_f = timeit(lambda: ops.depthwise_conv2d(x, w, None, s, p, d))
    t_all = timeit(lambda: torch.autograd.grad(ops.depthwise_conv2d(x, w, None, s, p, d), (x, w), dy))
    by = 2.0 * (x.numel() + y.numel())
    print(json.dumps({'layer': name, 'N': n, 'C': c, 'HW': h, 'k': k, 'fwd_us': round(t_f * 1e6, 1), 'fwd_GBps': round(by / t_f / 1e9),
                      'fwd_bwd_us': round(t_all * 1e6, 1), 'bwd_GBps_3_passes': round(2.0 * by / (t_all - t_f) / 1e9)}), flush=True)
