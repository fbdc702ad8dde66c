"""Generates tests/golden/backbone_<name>.pt by RUNNING THE REFERENCE classification backbones (imported from /root/reference) that
reuse the hot-path blocks: darknettiny / darknet19 / darknet53 (backbones/darknet.py), van_b0
(backbones/van.py) and a small ConvFormer (backbones/convformer.py MetaFormer) -- fp32 on the CPU, one seeded batch, forward
logits, a fixed random projection of them back-propagated (per-parameter gradient norms + samples), BatchNorm buffers, and how far
the reference's OWN bf16-autocast run moves from its fp32 run (the gate of the product's bf16 test).

Build container only:   python oracle/make_golden_backbones.py [name ...]
The product-side test (tests/test_gpu_backbones.py) builds the same model under the same seed (identical initial weights, checked
on samples of every tensor) and applies the same `prepare` step."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')

# name -> (module, factory or class, kwargs, (batch, height, width))
CASES = {
    'darknettiny': ('darknet', 'darknettiny', dict(num_classes=24), (4, 96, 96)),
    'darknet19': ('darknet', 'darknet19', dict(num_classes=24), (4, 96, 96)),
    'darknet53': ('darknet', 'darknet53', dict(num_classes=24), (4, 96, 96)),
    'darknet19_silu': ('darknet', 'darknet19', dict(num_classes=24, act_type='silu'), (4, 96, 96)),
    'van_b0': ('van', 'van_b0', dict(num_classes=24), (2, 128, 128)),
    'convformer_tiny': ('convformer', 'MetaFormer', dict(num_classes=24, embedding_planes=[32, 64, 96, 128], block_nums=[1, 1, 2, 1]),
                        (2, 128, 128)),
}


def prepare(model):
    """VAN starts its residual branches at layer_scale 1e-5, which would hide them from an output comparison: 0.5 on both sides."""
    with torch.no_grad():
        for k, p in model.named_parameters():
            if 'layer_scale' in k:
                p.fill_(0.5)
    return model


def sample_idx(numel, k=16):
    return torch.linspace(0, numel - 1, min(k, numel)).long()


def build(name):
    import importlib
    mod, fn, kwargs, shape = CASES[name]
    m = importlib.import_module(f'SimpleAICV.classification.backbones.{mod}')
    torch.manual_seed(0)
    model = getattr(m, fn)(**kwargs)
    return model, kwargs, shape


def run(name):
    model, kwargs, (b, h, w) = build(name)
    init = {k: v.clone() for k, v in model.state_dict().items()}
    prepare(model).train()
    x = torch.randn(b, h, w, 3, generator=torch.Generator().manual_seed(1)).permute(0, 3, 1, 2)
    logits = model(x)
    proj = torch.randn(logits.shape, generator=torch.Generator().manual_seed(2))
    loss = (logits.float() * proj).sum() / logits.numel() ** 0.5
    loss.backward()
    fx = {'config': kwargs, 'input_shape': (b, 3, h, w), 'logits': logits.detach().clone(), 'scalar': float(loss),
          'init_sample': {k: v.flatten()[sample_idx(v.numel())].clone() for k, v in init.items() if v.dtype.is_floating_point},
          'grad_norm': {k: float(p.grad.norm()) for k, p in model.named_parameters() if p.grad is not None},
          'grad_sample': {k: p.grad.flatten()[sample_idx(p.numel())].clone() for k, p in model.named_parameters() if p.grad is not None},
          'bn_buffers': {k: v.clone() for k, v in list(model.state_dict().items()) if 'running_' in k}}
    model2, _, _ = build(name)
    prepare(model2).train()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        l2 = model2(x)
    fx['bf16_dev'] = float((l2.float() - logits).abs().max() / logits.abs().max())
    torch.save(fx, os.path.join(OUT, f'backbone_{name}.pt'))
    print(name, tuple(logits.shape), 'scalar', round(float(loss), 5), 'params', len(fx['grad_norm']), 'bf16 dev', round(fx['bf16_dev'], 4),
          'bytes', os.path.getsize(os.path.join(OUT, f'backbone_{name}.pt')))


def main():
    sys.path.insert(0, REF)
    import types
    if 'calflops' not in sys.modules:
        sys.modules['calflops'] = types.ModuleType('calflops')
    for name in (sys.argv[1:] or list(CASES)):
        run(name)


if __name__ == '__main__':
    main()
