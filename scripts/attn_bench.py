"""Streaming-attention microbench on one MI355X: SAM global block (N = 4096, rel-pos), SAM window (N = 196),
DETR encoder (N = 1764, head dim 32, key bias)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleaicv_pytorch_training_examples_amd import ops_tfm  # noqa: E402
from scripts.kernel_bench import timeit  # noqa: E402


def run(name, b, heads, d, n, rel, kb):
    c = heads * d
    qkv = torch.randn(b, n, 3 * c, device='cuda').bfloat16()
    q, k, v = qkv[:, :, :c], qkv[:, :, c:2 * c], qkv[:, :, 2 * c:]
    rel_h = rel_w = key_bias = None
    if rel:
        rel_h = torch.randn(b * heads, n, rel[0], device='cuda')
        rel_w = torch.randn(b * heads, n, rel[1], device='cuda')
    if kb:
        key_bias = (torch.rand(b, n, device='cuda') > 0.7).float()
    scale = d ** -0.5
    out, lse = ops_tfm.sattn_fwd(q, k, v, heads, scale, key_bias, rel_h, rel_w)
    dout = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :, :c], dqkv[:, :, c:2 * c], dqkv[:, :, 2 * c:]
    tf = timeit(lambda: ops_tfm.sattn_fwd(q, k, v, heads, scale, key_bias, rel_h, rel_w))
    tb = timeit(lambda: ops_tfm.sattn_bwd(q, k, v, out, dout, lse, heads, scale, dq, dk, dv, key_bias, rel_h, rel_w))
    fl = 4.0 * b * heads * n * n * d
    print(json.dumps({'case': name, 'fwd_us': round(tf * 1e6, 1), 'fwd_tf': round(fl / tf / 1e12, 1),
                      'bwd_us': round(tb * 1e6, 1), 'bwd_tf': round(2.5 * fl / tb / 1e12, 1)}), flush=True)


CASES = [('sam_global_b8', 8, 12, 64, 4096, (64, 64), False), ('sam_window_b8', 200, 12, 64, 196, (14, 14), False),
         ('detr_enc_b8', 8, 8, 32, 1764, None, True), ('plain_d64_n4096_b8', 8, 12, 64, 4096, None, False),
         ('vit_b256_n197', 256, 12, 64, 197, None, False), ('sam_window_norel_b8', 200, 12, 64, 196, None, False)]
ONLY = [c for c in os.environ.get('ATTN_CASES', '').split(',') if c]
for case in CASES:
    if not ONLY or case[0] in ONLY:
        run(*case)
