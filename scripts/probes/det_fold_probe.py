"""r06: the deterministic fold's forms (csrc/det.hip).  SAICV_ORDERED_FOLD=chain: one thread per element quad walks every part; default:
statistics folds through the cooperative kernel (loads spread over eight lanes, additions in the chain's order: must be BIT-EQUAL to
the chain), weight-gradient folds through the eight-lane association.  Column sums and BatchNorm statistics with 32 ... 128 parts:
error against float64, bit-equality run to run and between the two settings; then the ResNet-50 bench line under each."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    from simpleaicv_pytorch_training_examples_amd import ops
    from simpleaicv_pytorch_training_examples_amd._lib import lib
    from simpleaicv_pytorch_training_examples_amd.ops import check, dtype_code, ptr, stream
    ops.set_deterministic(True)
    g = torch.Generator().manual_seed(5)
    out = {}
    for m, n in ((16384, 1024), (200000, 64), (4096, 4096)):
        x = (torch.randn(m, n, generator=g) + 0.25).cuda()
        ref = x.double().sum(0)
        res = []
        for _ in range(2):
            o = torch.zeros(n, device='cuda')
            check(lib().saicv_colsum(dtype_code(x.dtype), ptr(x), m, n, ptr(o), stream()), 'colsum')
            torch.cuda.synchronize()
            res.append(o.clone())
        out[f'{m}x{n}'] = {'rel_err_vs_f64': float(((res[0].double() - ref).abs() / ref.abs().clamp_min(1.0)).max()),
                           'bit_equal_rerun': bool(torch.equal(res[0], res[1])), 'bits': int(res[0].view(torch.int32).to(torch.int64).sum())}
        xb = x.to(torch.bfloat16)
        sm, sq = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
        check(lib().saicv_bn_stats(dtype_code(xb.dtype), ptr(xb), m, n, ptr(sm), ptr(sq), stream()), 'bn_stats')
        torch.cuda.synchronize()
        out[f'{m}x{n}']['bn_stats_bits'] = [int(sm.view(torch.int32).to(torch.int64).sum()), int(sq.view(torch.int32).to(torch.int64).sum())]
        out[f'{m}x{n}']['bn_stats_err'] = float(((sm.double() - xb.double().sum(0)).abs() / xb.double().sum(0).abs().clamp_min(1.0)).max())
    print(json.dumps({'fold': os.environ.get('SAICV_ORDERED_FOLD'), 'colsum': out}), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
        sys.exit(0)
    seen = {}
    for fold in ('chain', 'default'):
        env = dict(os.environ, SAICV_ORDERED_FOLD=fold)
        c = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, check=False, capture_output=True, text=True)
        line = [l for l in c.stdout.splitlines() if l.startswith('{')]
        print(line[-1] if line else c.stderr[-1200:], flush=True)
        seen[fold] = json.loads(line[-1])['colsum'] if line else None
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--model', 'resnet50', '--deterministic', '--no-secondary', '--no-cpu-baseline',
                            '--no-sam', '--max-windows', '2'], env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(json.dumps({'fold': fold, 'ms_per_step': d['ms_per_step'], 'final_loss': d['config']['final_loss'],
                              'bn_statistics': d['config']['bn_statistics']}), flush=True)
        except Exception as e:                                  # noqa: BLE001
            print('bench failed', fold, e, r.stderr[-800:], flush=True)
    print('statistics folds bit-equal between chain and default:', seen['chain'] is not None and seen['chain'] == seen['default'], flush=True)
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--model', 'resnet50', '--no-secondary', '--no-cpu-baseline', '--no-sam',
                        '--max-windows', '2', '--steps', '20'], env=env, capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps({'fold': 'fast mode', 'ms_per_step': d['ms_per_step'], 'final_loss': d['config']['final_loss']}), flush=True)
