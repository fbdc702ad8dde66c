"""r06: how far do the detection VAN / ConvFormer fp32 gradient norms sit from the reference's under different (all fixed, all
bit-reproducible) associations of the deterministic fold?  SAICV_ORDERED_FOLD = chain (one thread walks the parts in ascending order)
against the default (cooperative chain, weight gradients through the eight-lane form).  profiles/r06_van_fold_probe.txt also holds two
legs of an experiment build that is not kept: "all" (every fold with >= 32 parts through the eight lanes) and "rev" (the chain walked
from the last part) -- the legs that showed VAN's BatchNorm backward amplifying a re-association to 2e-4 / 1e-3."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def child(case):
    import torch
    from test_gpu_backbones import _det_backbone
    from simpleaicv_pytorch_training_examples_amd import ops
    ops.set_deterministic(True)
    fx, m, x, g = _det_backbone(case)
    m = m.cuda().train()
    outs = m(x.cuda())
    probes = [torch.randn(sh, generator=g) for sh in fx['out_shapes']]
    sum((o.float() * p.cuda()).sum() for o, p in zip(outs, probes)).backward()
    torch.cuda.synchronize()
    top = max(fx['grad_norm'].values())
    errs = sorted(((abs(float(p.grad.float().norm()) - fx['grad_norm'][n]) / fx['grad_norm'][n], n) for n, p in m.named_parameters()
                   if fx['grad_norm'][n] > 1e-6 * top), reverse=True)
    oerr = [abs(float(o.float().norm()) - fx['out_norm'][i]) / fx['out_norm'][i] for i, o in enumerate(outs)]
    print(f"{case:16s} fold={os.environ.get('SAICV_ORDERED_FOLD'):6s} outputs {max(oerr):.1e}  worst gradient norms: " +
          ', '.join(f'{n} {e:.1e}' for e, n in errs[:4]) + f'   median {errs[len(errs) // 2][0]:.1e}', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'child':
        child(sys.argv[2])
        sys.exit(0)
    for case in ('van', 'convformer', 'dinov3convnext'):
        for fold in ('chain', 'default'):
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child', case], env=dict(os.environ, SAICV_ORDERED_FOLD=fold))
