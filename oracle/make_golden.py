"""Generates tests/golden/*.pt by RUNNING THE REFERENCE ITSELF (imported from /root/reference).

Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden.py

Each fixture holds: the model name + ctor kwargs, the seeds, the CPU-generated input batch
checksum, reference logits, loss, per-parameter gradient norms, a 64-element sample of every
gradient, the full gradient of a few small parameters, and the BN running statistics after one
training forward.  Inputs are regenerated from the seed by tests (same torch build on both
machines); the checksum guards that.
"""
import os
import sys

import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def make_batch(seed, shape, num_classes, soft=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    if soft:
        a = torch.randint(0, num_classes, (shape[0],), generator=g)
        b = torch.randint(0, num_classes, (shape[0],), generator=g)
        lam = 0.7
        ya = torch.full((shape[0], num_classes), 0.1 / num_classes).scatter_(1, a[:, None], 1 - 0.1 + 0.1 / num_classes)
        yb = torch.full((shape[0], num_classes), 0.1 / num_classes).scatter_(1, b[:, None], 1 - 0.1 + 0.1 / num_classes)
        y = lam * ya + (1 - lam) * yb
    else:
        y = torch.randint(0, num_classes, (shape[0],), generator=g)
    return x, y


def grad_summary(model):
    norms, samples, full = {}, {}, {}
    for n, p in model.named_parameters():
        g = p.grad.detach()
        norms[n] = float(g.norm())
        samples[n] = g.flatten()[:64].clone()
        if g.numel() <= 4096:
            full[n] = g.clone()
    return norms, samples, full


def run_case(name, factory, kwargs, shape, num_classes, criterion, soft, model_seed=0, data_seed=1):
    torch.manual_seed(model_seed)
    model = factory(**kwargs)
    model.train()
    x, y = make_batch(data_seed, shape, num_classes, soft)
    logits = model(x)
    loss = criterion(logits, y)
    loss.backward()
    norms, samples, full = grad_summary(model)
    buffers = {n: b.detach().clone() for n, b in model.named_buffers() if b.numel() <= 4096}
    # eval-mode logits with the updated running stats
    model.eval()
    with torch.no_grad():
        eval_logits = model(x)
    fx = {
        'name': name, 'kwargs': kwargs, 'shape': list(shape), 'num_classes': num_classes, 'soft': soft,
        'model_seed': model_seed, 'data_seed': data_seed,
        'input_checksum': float(x.double().sum()), 'label_checksum': float(y.double().sum()),
        'logits': logits.detach().clone(), 'loss': float(loss), 'eval_logits': eval_logits.clone(),
        'grad_norm': norms, 'grad_sample': samples, 'grad_full': full, 'buffers_after': buffers,
        'torch_version': torch.__version__,
    }
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.pt')
    torch.save(fx, path)
    print(f'{name}: loss={float(loss):.6f} logits_norm={float(logits.norm()):.5f} -> {path} '
          f'({os.path.getsize(path) / 1024:.0f} KiB)')


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    from SimpleAICV.classification import backbones, losses   # the reference's own modules

    ce, soft_ce = losses.CELoss(), losses.OneHotLabelCELoss()
    run_case('resnet18cifar_b8', backbones.resnet18cifar, {'num_classes': 100}, (8, 3, 32, 32), 100, ce, False)
    run_case('resnet50_b4_64', backbones.resnet50, {'num_classes': 1000}, (4, 3, 64, 64), 1000, ce, False)
    run_case('resnet50_b2_224', backbones.resnet50, {'num_classes': 1000}, (2, 3, 224, 224), 1000, ce, False)
    run_case('resnet34_b2_96', backbones.resnet34, {'num_classes': 10}, (2, 3, 96, 96), 10, ce, False)
    run_case('vit_base_patch16_b2_224', backbones.vit_base_patch16,
             {'image_size': 224, 'drop_path_prob': 0.0, 'global_pool': True, 'num_classes': 1000},
             (2, 3, 224, 224), 1000, soft_ce, True)
    run_case('vit_tiny_b3_64', lambda **kw: backbones.vit._vit(16, 192, 3, 3, 4, **kw),
             {'image_size': 64, 'drop_path_prob': 0.0, 'global_pool': False, 'num_classes': 10},
             (3, 3, 64, 64), 10, ce, False)


if __name__ == '__main__':
    main()
