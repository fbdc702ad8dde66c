"""ViT-B MLP / attention-projection GEMMs WITH the epilogues they carry inside the model (M = 197 * 256 rows, bf16): plain,
bias + residual + drop-path row scale (proj / fc2 forward), GELU with two outputs (fc1 forward, saicv_linear_gelu_fwd_aux) and
the stored-derivative multiply (fc2 data gradient, saicv_linear_dgrad_mul).  The in-model profile (r04a) prices the two GELU
epilogues at +83 / +67 us per layer over the plain GEMMs: this is the A/B harness for the epilogue / stagger work.
Env: LF_ENVS="A=1,B=2;C=3" runs the table once per ';'-separated variant."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from simpleaicv_pytorch_training_examples_amd import _lib  # noqa: E402
from simpleaicv_pytorch_training_examples_amd._lib import check, lib, ptr  # noqa: E402
from kernel_bench import timeit  # noqa: E402


def run(M, tag):
    L, st = lib(), _lib.stream()
    bf = torch.bfloat16
    C, H = 768, 3072
    x = torch.randn(M, C, device='cuda').to(bf)
    h = torch.randn(M, H, device='cuda').to(bf)
    w_qkv = (torch.randn(3 * C, C, device='cuda') * 0.03).to(bf)
    w_proj = (torch.randn(C, C, device='cuda') * 0.03).to(bf)
    w_fc1 = (torch.randn(H, C, device='cuda') * 0.03).to(bf)
    w_fc2 = (torch.randn(C, H, device='cuda') * 0.03).to(bf)
    b_c = torch.randn(C, device='cuda')
    b_h = torch.randn(H, device='cuda')
    b_q = torch.randn(3 * C, device='cuda')
    y_c = torch.empty(M, C, device='cuda', dtype=bf)
    y_h = torch.empty(M, H, device='cuda', dtype=bf)
    y_h2 = torch.empty(M, H, device='cuda', dtype=bf)
    y_q = torch.empty(M, 3 * C, device='cuda', dtype=bf)
    res = torch.randn(M, C, device='cuda').to(bf)
    scale = (torch.rand(M // 197, device='cuda') > 0.1).float() / 0.9
    fac = torch.rand(M, H, device='cuda').to(bf)
    rows = []

    def rec(name, fn, flops):
        t = timeit(fn)
        rows.append((name, t))
        print(json.dumps({'tag': tag, 'gemm': name, 'us': round(t * 1e6, 1), 'TFLOPs': round(flops / t / 1e12, 1)}), flush=True)

    f_qkv, f_proj, f_mlp = 2.0 * M * C * 3 * C, 2.0 * M * C * C, 2.0 * M * C * H
    rec('qkv fwd (bias)', lambda: check(L.saicv_linear_fwd(0, ptr(x), ptr(w_qkv), ptr(b_q), ptr(y_q), M, C, 3 * C, 0, 0, 0, 1, st)), f_qkv)
    rec('proj fwd (bias + residual + row scale)', lambda: check(L.saicv_linear_fwd(0, ptr(x), ptr(w_proj), ptr(b_c), ptr(y_c), M, C, C, 0, ptr(res), ptr(scale), 197, st)), f_proj)
    rec('fc1 fwd plain (bias)', lambda: check(L.saicv_linear_fwd(0, ptr(x), ptr(w_fc1), ptr(b_h), ptr(y_h), M, C, H, 0, 0, 0, 1, st)), f_mlp)
    rec('fc1 fwd gelu, two outputs', lambda: check(L.saicv_linear_gelu_fwd_aux(0, ptr(x), ptr(w_fc1), ptr(b_h), ptr(y_h), ptr(y_h2), M, C, H, st)), f_mlp)
    rec('fc2 fwd (bias + residual + row scale)', lambda: check(L.saicv_linear_fwd(0, ptr(h), ptr(w_fc2), ptr(b_c), ptr(y_c), M, H, C, 0, ptr(res), ptr(scale), 197, st)), f_mlp)
    # data gradients: dx[M][K] = dy[M][N] wd[K][N]^T
    wd_fc2 = (torch.randn(H, C, device='cuda') * 0.03).to(bf)          # fc2: K = 3072 (dx), N = 768 (dy)
    wd_fc1 = (torch.randn(C, H, device='cuda') * 0.03).to(bf)          # fc1: K = 768, N = 3072
    wd_qkv = (torch.randn(C, 3 * C, device='cuda') * 0.03).to(bf)
    rec('fc2 dgrad plain', lambda: check(L.saicv_linear_dgrad(0, ptr(x), ptr(wd_fc2), ptr(y_h), M, H, C, 0, st)), f_mlp)
    rec('fc2 dgrad x stored gelu\'', lambda: check(L.saicv_linear_dgrad_mul(0, ptr(x), ptr(wd_fc2), ptr(fac), ptr(y_h), M, H, C, st)), f_mlp)
    rec('fc1 dgrad plain', lambda: check(L.saicv_linear_dgrad(0, ptr(h), ptr(wd_fc1), ptr(y_c), M, C, H, 0, st)), f_mlp)
    rec('qkv dgrad plain', lambda: check(L.saicv_linear_dgrad(0, ptr(y_q), ptr(wd_qkv), ptr(y_c), M, C, 3 * C, 0, st)), f_qkv)
    rec('proj dgrad plain', lambda: check(L.saicv_linear_dgrad(0, ptr(x), ptr(w_proj), ptr(y_c), M, C, C, 0, st)), f_proj)
    # one ViT-B layer's forward + data-gradient GEMMs as the model runs them
    layer = sum(t for n, t in rows if n in ('qkv fwd (bias)', 'proj fwd (bias + residual + row scale)', 'fc1 fwd gelu, two outputs',
                                              'fc2 fwd (bias + residual + row scale)', 'fc2 dgrad x stored gelu\'', 'fc1 dgrad plain',
                                              'qkv dgrad plain', 'proj dgrad plain'))
    print(json.dumps({'tag': tag, 'vit_b_layer_nt_us': round(layer * 1e6, 1), 'x12_ms': round(layer * 12e3, 3)}), flush=True)


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
    variants = [v for v in os.environ.get('LF_ENVS', '').split(';') if v] or ['']
    for v in variants:
        sets = dict(kv.split('=') for kv in v.split(',') if kv)
        for k_, val in sets.items():
            os.environ[k_] = val
        run(M, v or 'default')
        for k_ in sets:
            os.environ.pop(k_, None)


if __name__ == '__main__':
    main()
