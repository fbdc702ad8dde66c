"""Per-kernel microbenchmark on one MI355X: every distinct ResNet-50 conv shape (fwd / dgrad /
wgrad, bf16) against the MFMA roofline and the BN / pooling streaming kernels against the HBM
roofline.  Writes JSON lines to stdout; used to pick optimisation targets (profiles/)."""
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

# (Cin, Cout, k, stride, Hin) distinct ResNet-50 convs at 224x224 input
R50 = [(8, 64, 7, 2, 224), (64, 64, 1, 1, 56), (64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56),
       (256, 128, 1, 1, 56), (128, 128, 3, 2, 56), (128, 512, 1, 1, 28), (256, 512, 1, 2, 56), (512, 128, 1, 1, 28),
       (128, 128, 3, 1, 28), (512, 256, 1, 1, 28), (256, 256, 3, 2, 28), (256, 1024, 1, 1, 14), (512, 1024, 1, 2, 28),
       (1024, 256, 1, 1, 14), (256, 256, 3, 1, 14), (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14), (512, 2048, 1, 1, 7),
       (1024, 2048, 1, 2, 14), (2048, 512, 1, 1, 7), (512, 512, 3, 1, 7)]


ITERS = int(os.environ.get('KB_ITERS', '10'))
PEAK_F, PEAK_B = 2.5e15, 8.0e12        # dense bf16 MFMA, HBM3E (MI355X_MICROARCH.md)


def multiplicity():
    """How many times each distinct shape occurs in ResNet-50 (stem counted as its 8-channel packed form)."""
    from r50_roofline import resnet50_convs
    n = {}
    for ci, co, k, s, h in resnet50_convs():
        key = (8 if ci == 3 else ci, co, k, s, h)
        n[key] = n.get(key, 0) + 1
    return n


def bounds(batch, ci, co, k, s, h, oh):
    """Lower bound (s) of each leg = max(flops / MFMA peak, algorithmic bytes / HBM peak); scripts/r50_roofline.py."""
    cin = 3 if ci == 8 else ci
    flops = 2.0 * batch * oh * oh * co * cin * k * k
    xin = batch * (oh * oh if (k == 1 and s == 2) else h * h) * ci * 2
    yout = batch * oh * oh * co * 2
    w = co * ci * k * k
    by = {'fwd': xin + w * 2 + yout, 'dgrad': yout + w * 2 + batch * h * h * ci * 2, 'wgrad': xin + yout + w * 4}
    return {leg: max(flops / PEAK_F, b / PEAK_B) for leg, b in by.items()}


def timeit(fn, iters=None, warm=2):
    iters = iters or ITERS
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dt = torch.bfloat16
    L = lib()
    st = _lib.stream()
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0, 'flops': 0.0}
    mult = multiplicity()
    model = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    model_bound = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    only = [int(i) for i in os.environ['KB_ONLY'].split(',')] if os.environ.get('KB_ONLY') else None
    for idx, (ci, co, k, s, h) in enumerate(R50):
        if only is not None and idx not in only:
            continue
        pad = k // 2
        d = ops._desc(batch, h, h, ci, co, k, k, s, pad, dt)
        x = torch.randn(batch, h, h, ci, device='cuda').to(dt)
        wf = (torch.randn(co, k, k, ci, device='cuda') * 0.05).to(dt)
        wd = (torch.randn(ci, k, k, co, device='cuda') * 0.05).to(dt)
        y = torch.empty(batch, d.OH, d.OW, co, device='cuda', dtype=dt)
        dy = torch.randn(batch, d.OH, d.OW, co, device='cuda').to(dt)
        dx = torch.empty_like(x)
        dw = torch.zeros(co, k, k, ci, device='cuda')
        rows = L.saicv_conv2d_stat_rows(ctypes.byref(d))
        stats = torch.empty(2, rows, co, device='cuda')
        flops = 2.0 * batch * d.OH * d.OW * co * k * k * ci
        nostat = bool(os.environ.get('KB_NO_STATS'))
        t_f = timeit(lambda: check(L.saicv_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(wf), 0, ptr(y), 0, 0 if nostat else ptr(stats[0]), 0 if nostat else ptr(stats[1]), st)))
        t_d = timeit(lambda: check(L.saicv_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dx), st))) if ci != 8 else 0.0
        t_w = timeit(lambda: check(L.saicv_conv2d_wgrad(ctypes.byref(d), ptr(dy), ptr(x), ptr(dw), st))) if not os.environ.get('KB_SKIP_WGRAD') else 1.0
        rec = {'conv': f'{ci}->{co} k{k} s{s} {h}->{d.OH}', 'gflop': round(flops / 1e9, 2),
               'fwd_us': round(t_f * 1e6, 1), 'fwd_tflops': round(flops / t_f / 1e12, 1),
               'dgrad_us': round(t_d * 1e6, 1), 'dgrad_tflops': round(flops / t_d / 1e12, 1) if t_d else None,
               'wgrad_us': round(t_w * 1e6, 1), 'wgrad_tflops': round(flops / t_w / 1e12, 1)}
        lb = bounds(batch, ci, co, k, s, h, d.OH)
        n = mult.get((ci, co, k, s, h), 1)
        rec['count'] = n
        for leg, t in (('fwd', t_f), ('dgrad', t_d), ('wgrad', t_w)):
            if t:
                rec[leg + '_bound_us'] = round(lb[leg] * 1e6, 1)
                rec[leg + '_frac_of_bound'] = round(lb[leg] / t, 3)
                model[leg] += n * t
                model_bound[leg] += n * lb[leg]
        print(json.dumps(rec), flush=True)
        tot['fwd'] += t_f
        tot['dgrad'] += t_d
        tot['wgrad'] += t_w
        tot['flops'] += flops
        del x, wf, wd, y, dy, dx, dw
    print(json.dumps({'distinct_shape_totals_ms': {k: round(v * 1e3, 3) for k, v in tot.items() if k != 'flops'}}))
    print(json.dumps({'resnet50_all_53_convs_ms': {k: round(v * 1e3, 3) for k, v in model.items()},
                      'lower_bound_ms': {k: round(v * 1e3, 3) for k, v in model_bound.items()},
                      'frac_of_bound': {k: round(model_bound[k] / v, 3) for k, v in model.items() if v}}))

    if os.environ.get('KB_CONV_ONLY'):
        return
    # streaming kernels on the largest activation (layer1 output: [B,56,56,256])
    M, C = batch * 56 * 56, 256
    yb = torch.randn(M, C, device='cuda').to(dt)
    rb = torch.randn(M, C, device='cuda').to(dt)
    zb = torch.empty_like(yb)
    dzb = torch.randn(M, C, device='cuda').to(dt)
    dyb = torch.empty_like(yb)
    drb = torch.empty_like(yb)
    sc = torch.rand(C, device='cuda') + 0.5
    sh = torch.randn(C, device='cuda')
    mean = torch.randn(C, device='cuda') * 0.1
    invstd = torch.rand(C, device='cuda') + 0.5
    dg = torch.empty(C, device='cuda')
    db = torch.empty(C, device='cuda')
    ws = torch.empty(L.saicv_bn_bwd_ws_floats(M, C, 0), device='cuda')
    mk = torch.empty(M * C // 8, dtype=torch.uint8, device='cuda')
    t = timeit(lambda: check(L.saicv_bn_act_fwd(0, ptr(yb), ptr(rb), ptr(zb), ptr(sc), ptr(sh), M, C, 1, ptr(mk), st)))
    by = M * C * 2 * 3
    print(json.dumps({'kernel': 'bn_act_fwd(+res,relu)', 'us': round(t * 1e6, 1), 'GBps': round(by / t / 1e9, 1), 'frac_8TBps': round(by / t / 8e12, 3)}))
    t = timeit(lambda: check(L.saicv_bn_act_bwd(0, ptr(dzb), 0, ptr(mk), ptr(yb), ptr(sc), ptr(mean), ptr(invstd), ptr(dyb), ptr(drb), ptr(dg), ptr(db), M, C, 1, 0, ptr(ws), st)))
    by = M * C * 2 * (2 + 2 + 2) + 2 * M * C // 8
    print(json.dumps({'kernel': 'bn_act_bwd(+res,relu)', 'us': round(t * 1e6, 1), 'GBps': round(by / t / 1e9, 1), 'frac_8TBps': round(by / t / 8e12, 3)}))


if __name__ == '__main__':
    main()
