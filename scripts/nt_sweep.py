"""Implicit-GEMM forward / data-gradient sweep on one MI355X: every tile geometry (SAICV_NT_TILE) with and without the
persistent tile loop (SAICV_NT_PERSIST) over the four ViT-B GEMM shapes, two large square ones and the 23 distinct ResNet-50
convolutions at batch 256, in ONE process.  Prints JSON lines: per shape and configuration the time, TFLOP/s and the largest
deviation of the output from the default configuration's (every geometry runs the same arithmetic: they must agree to bf16
rounding of identical fp32 sums, i.e. exactly, except where split points of fp32 accumulation differ -- they do not here)."""
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
from kernel_bench import R50, timeit  # noqa: E402

# (name, SAICV_NT_TILE, SAICV_NT_PERSIST, SAICV_NT_STAGGER)
CONFIGS = [('auto_np', None, '0', '0'), ('auto', None, '1', '4'), ('t0_np', '0', '0', '0'), ('t0_s0', '0', '1', '0'), ('t0_s2', '0', '1', '2'),
           ('t0_s4', '0', '1', '4'), ('t0_s8', '0', '1', '8'), ('t4_s0', '4', '1', '0'), ('t4_s2', '4', '1', '2'), ('t4_s4', '4', '1', '4'),
           ('t4_s8', '4', '1', '8'), ('t2_s4', '2', '1', '4')]


def setcfg(tile, persist, stagger='0'):
    if tile is None:
        os.environ.pop('SAICV_NT_TILE', None)
    else:
        os.environ['SAICV_NT_TILE'] = tile
    os.environ['SAICV_NT_PERSIST'] = persist
    os.environ['SAICV_NT_STAGGER'] = stagger


def main():
    L, st = lib(), _lib.stream()
    dt = torch.bfloat16
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    M0 = 50432
    if which in ('all', 'linear'):
        for (M, K, N) in [(M0, 768, 2304), (M0, 768, 768), (M0, 768, 3072), (M0, 3072, 768), (4096, 4096, 4096), (8192, 8192, 8192)]:
            x = torch.randn(M, K, device='cuda').to(dt)
            wf = (torch.randn(N, K, device='cuda') * 0.03).to(dt)
            wd = wf.t().contiguous()
            dy = torch.randn(M, N, device='cuda').to(dt)
            bias = torch.randn(N, device='cuda')
            fl = 2.0 * M * K * N
            ref = None
            rec = {'gemm': f'{M}x{K}x{N}'}
            for name, tile, persist, stag in CONFIGS:
                setcfg(tile, persist, stag)
                y = torch.empty(M, N, device='cuda', dtype=dt)
                dx = torch.empty(M, K, device='cuda', dtype=dt)
                tf = timeit(lambda: check(L.saicv_linear_fwd(0, ptr(x), ptr(wf), ptr(bias), ptr(y), M, K, N, 0, 0, 0, 1, st)))
                td = timeit(lambda: check(L.saicv_linear_dgrad(0, ptr(dy), ptr(wd), ptr(dx), M, K, N, 0, st)))
                torch.cuda.synchronize()
                if ref is None:
                    ref = (y.float().clone(), dx.float().clone())
                err = max(float((y.float() - ref[0]).abs().max()), float((dx.float() - ref[1]).abs().max()))
                rec[name] = [round(fl / tf / 1e12), round(fl / td / 1e12), err]
                del y, dx
            print(json.dumps(rec), flush=True)
            del x, wf, wd, dy
    if which in ('all', 'conv'):
        batch = 256
        tot = {}
        for idx, (ci, co, k, s, h) in enumerate(R50):
            pad = k // 2
            d = ops._desc(batch, h, h, ci, co, k, k, s, pad, dt)
            x = torch.randn(batch, h, h, ci, device='cuda').to(dt)
            wf = (torch.randn(co, k, k, ci, device='cuda') * 0.05).to(dt)
            wd = (torch.randn(ci, k, k, co, device='cuda') * 0.05).to(dt)
            dy = torch.randn(batch, d.OH, d.OW, co, device='cuda').to(dt)
            fl = 2.0 * batch * d.OH * d.OW * co * k * k * ci
            rec = {'conv': f'{ci}->{co} k{k} s{s} {h}'}
            ref = None
            for name, tile, persist, stag in CONFIGS:
                setcfg(tile, persist, stag)
                y = torch.empty(batch, d.OH, d.OW, co, device='cuda', dtype=dt)
                dx = torch.empty_like(x)
                rows = L.saicv_conv2d_stat_rows(ctypes.byref(d))
                stats = torch.zeros(2, rows, co, device='cuda')
                tf = timeit(lambda: check(L.saicv_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(wf), 0, ptr(y), 0, ptr(stats[0]), ptr(stats[1]), st)))
                td = timeit(lambda: check(L.saicv_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dx), st))) if ci != 8 else 1.0
                torch.cuda.synchronize()
                ssum = stats[0].sum(0).clone()          # (rows are OVERWRITTEN by each launch in this mode)
                if ref is None:
                    ref = (y.float().clone(), dx.float().clone(), ssum)
                err = max(float((y.float() - ref[0]).abs().max()), float((dx.float() - ref[1]).abs().max()) if ci != 8 else 0.0)
                serr = float((ssum - ref[2]).abs().max() / ref[2].abs().max().clamp_min(1e-6))
                rec[name] = [round(tf * 1e6, 1), round(td * 1e6, 1), err, round(serr, 6)]
                t = tot.setdefault(name, [0.0, 0.0])
                t[0] += tf
                t[1] += td if ci != 8 else 0.0
                del y, dx, stats
            print(json.dumps(rec), flush=True)
            del x, wf, wd, dy
        print(json.dumps({'distinct_conv_totals_ms': {k: [round(v[0] * 1e3, 3), round(v[1] * 1e3, 3)] for k, v in tot.items()}}))


if __name__ == '__main__':
    main()
