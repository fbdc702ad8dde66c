"""Data-gradient launches of ResNet-50 (b256, bf16) WITH the fused epilogue they carry inside the model (gated shortcut
gradient + BatchNorm-backward partial sums, saicv_conv2d_dgrad_fused) next to the plain data gradient of the same shape, each
against its HBM bound (algorithmic bytes / 6.3 TB/s achievable, / 8 TB/s peak).  The in-model profile (r04a) shows these
launches at 0.55-0.7 of that bound: this bench is the A/B harness for the epilogue work.
Env: DG_ENVS="A=1,B=2;C=3" runs the whole table once per ';'-separated variant (variables set per call, the library reads
SAICV_NT_* per launch)."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))

from simpleaicv_pytorch_training_examples_amd import _lib, ops  # noqa: E402
from simpleaicv_pytorch_training_examples_amd._lib import check, lib, ptr  # noqa: E402
from kernel_bench import timeit  # noqa: E402

# (name, Cin, Cout, k, stride, Hin, addend?) -- the data gradient produces dx [B, Hin, Hin, Cin]
CASES = [
    ('conv1 l1 64<-256', 256, 64, 1, 1, 56, True), ('conv1 l2 128<-512', 512, 128, 1, 1, 28, True),
    ('conv1 l3 256<-1024', 1024, 256, 1, 1, 14, True), ('conv1 l4 512<-2048', 2048, 512, 1, 1, 7, True),
    ('conv3 l1 256<-64', 64, 256, 1, 1, 56, False), ('conv3 l2 512<-128', 128, 512, 1, 1, 28, False),
    ('conv3 l3 1024<-256', 256, 1024, 1, 1, 14, False), ('conv3 l4 2048<-512', 512, 2048, 1, 1, 7, False),
    ('conv2 l1 3x3 64', 64, 64, 3, 1, 56, False), ('conv2 l2 3x3 128', 128, 128, 3, 1, 28, False),
    ('conv2 l3 3x3 256', 256, 256, 3, 1, 14, False), ('conv2 l4 3x3 512', 512, 512, 3, 1, 7, False),
]


def run(batch, tag):
    dt = torch.bfloat16
    L, st = lib(), _lib.stream()
    out = []
    tot_p = tot_f = tot_b = 0.0
    for name, ci, co, k, s, h, has_add in CASES:
        pad = k // 2
        d = ops._desc(batch, h, h, ci, co, k, k, s, pad, dt)
        M_in = batch * h * h
        dy = torch.randn(batch, d.OH, d.OW, co, device='cuda').to(dt)
        wd = (torch.randn(ci, k, k, co, device='cuda') * 0.05).to(dt)
        dx = torch.empty(batch, h, h, ci, device='cuda', dtype=dt)
        addend = torch.randn(batch, h, h, ci, device='cuda').to(dt)
        y = torch.randn(batch, h, h, ci, device='cuda').to(dt)
        gate = torch.randint(0, 256, (M_in * ci // 8,), device='cuda', dtype=torch.uint8)
        mask = torch.randint(0, 256, (M_in * ci // 8,), device='cuda', dtype=torch.uint8)
        mean = torch.randn(ci, device='cuda')
        invstd = torch.rand(ci, device='cuda') + 0.5
        rows = ops._stat_rows(L.saicv_conv2d_dgrad_stat_rows(ctypes.byref(d)))
        part = torch.zeros(2, rows, ci, device='cuda')
        fuse = _lib.DgradFuse()
        if has_add:
            fuse.addend, fuse.addend_gate = ptr(addend), ptr(gate)
        fuse.bn_y, fuse.bn_mask, fuse.bn_mean, fuse.bn_invstd = ptr(y), ptr(mask), ptr(mean), ptr(invstd)
        fuse.part_g, fuse.part_gx, fuse.part_rows = ptr(part[0]), ptr(part[1]), rows
        t_p = timeit(lambda: check(L.saicv_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dx), st)))
        t_f = timeit(lambda: check(L.saicv_conv2d_dgrad_fused(ctypes.byref(d), ptr(dy), ptr(wd), ctypes.byref(fuse), ptr(dx), st)))
        es = 2
        by_plain = dy.numel() * es + wd.numel() * es + dx.numel() * es
        by_fused = by_plain + dx.numel() * es * (2 if has_add else 1) + dx.numel() / 8 * (2 if has_add else 1)
        flops = 2.0 * batch * d.OH * d.OW * co * k * k * ci
        bound = max(flops / 2.5e15, by_fused / 6.3e12)
        rec = {'tag': tag, 'case': name, 'plain_us': round(t_p * 1e6, 1), 'fused_us': round(t_f * 1e6, 1),
               'fused_bound_us': round(bound * 1e6, 1), 'fused_frac': round(bound / t_f, 3),
               'fused_TBps': round(by_fused / t_f / 1e12, 2), 'TFLOPs': round(flops / t_f / 1e12, 0)}
        print(json.dumps(rec), flush=True)
        n = {'conv1 l1': 3, 'conv1 l2': 4, 'conv1 l3': 6, 'conv1 l4': 3, 'conv3 l1': 3, 'conv3 l2': 4, 'conv3 l3': 6, 'conv3 l4': 3,
             'conv2 l1': 3, 'conv2 l2': 3, 'conv2 l3': 5, 'conv2 l4': 2}[name[:8]]
        tot_p += n * t_p
        tot_f += n * t_f
        tot_b += n * bound
        del dy, wd, dx, addend, y, gate, mask, part
    print(json.dumps({'tag': tag, 'model_weighted_ms': {'plain': round(tot_p * 1e3, 3), 'fused': round(tot_f * 1e3, 3),
                                                         'fused_bound': round(tot_b * 1e3, 3)}}), flush=True)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    variants = [v for v in os.environ.get('DG_ENVS', '').split(';') if v] or ['']
    for v in variants:
        sets = dict(kv.split('=') for kv in v.split(',') if kv)
        for k_, val in sets.items():
            os.environ[k_] = val
        run(batch, v or 'default')
        for k_ in sets:
            os.environ.pop(k_, None)


if __name__ == '__main__':
    main()
