"""Where does the run-to-run spread of VAN's fp32 gradient norms come from (2e-4 ... 1e-3 swings need a 1e4 x amplifier)?
Runs the detection VAN backbone fixture N times in the fast (atomic) mode and twice in deterministic mode; prints, per parameter in
BACKWARD order, the relative spread of its gradient norm and of its forward-side statistics (BatchNorm batch mean / var through the
running buffers), so that the first layer (from the loss) with a spread >> 1e-6 stands out."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from test_gpu_backbones import _det_backbone
    from simpleaicv_pytorch_training_examples_amd import ops
    case = sys.argv[1] if len(sys.argv) > 1 else 'van'
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4

    def once():
        fx, m, x, g = _det_backbone(case)
        m = m.cuda().train()
        outs = m(x.cuda())
        probes = [torch.randn(sh, generator=g) for sh in fx['out_shapes']]
        sum((o.float() * p.cuda()).sum() for o, p in zip(outs, probes)).backward()
        torch.cuda.synchronize()
        names = [n for n, _ in m.named_parameters()]
        return fx, names, {n: p.grad.double().cpu() for n, p in m.named_parameters()}, [o.detach().double().cpu() for o in outs], \
            {n: b.detach().double().cpu() for n, b in m.named_buffers() if b.dtype.is_floating_point}

    for mode in ('atomic', 'deterministic'):
        prev = ops.set_deterministic(mode == 'deterministic')
        res = [once() for _ in range(runs if mode == 'atomic' else 2)]
        ops.set_deterministic(prev)
        fx, names = res[0][0], res[0][1]
        print(f'==== {case} / {mode}: {len(res)} runs')
        for i in range(4):
            o = torch.stack([r[3][i] for r in res])
            print(f'  output {i}: spread {float((o.max(0).values - o.min(0).values).max() / o[0].abs().max()):.2e}')
        worst = []
        for n in names:
            gs = torch.stack([r[2][n] for r in res])
            norms = gs.flatten(1).norm(dim=1)
            ref_n = fx['grad_norm'][n]
            spread = float((norms.max() - norms.min()) / max(float(norms.mean()), 1e-30))
            elem = float((gs.max(0).values - gs.min(0).values).max() / max(float(gs[0].abs().max()), 1e-30))
            err = float(abs(norms[0] - ref_n) / max(ref_n, 1e-30))
            worst.append((spread, elem, err, n, float(norms.mean())))
        for spread, elem, err, n, mean in worst:
            flag = ' <<<' if spread > 1e-5 else ''
            print(f'  {n:70s} |g| {mean:.3e}  norm spread {spread:.1e}  element spread {elem:.1e}  vs reference {err:.1e}{flag}')
        bufs = res[0][4].keys()
        for n in bufs:
            bs = torch.stack([r[4][n] for r in res])
            sp = float((bs.max(0).values - bs.min(0).values).max() / max(float(bs[0].abs().max()), 1e-30))
            if sp > 1e-6:
                print(f'  buffer {n}: spread {sp:.1e}')


if __name__ == '__main__':
    main()
