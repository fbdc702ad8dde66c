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


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def measure_reference_noise(factory, kwargs, x, y, criterion, model_seed, samples, full, logits, loss):
    """How much the REFERENCE ITSELF moves under (a) a different fp32 summation order (its own
    modules run in channels_last on CPU) and (b) its own bf16 autocast (CPU).  Whole-model parity
    tolerances in tests/test_gpu_models.py are derived from these numbers: a deep BN network at
    batch 2-8 amplifies rounding differences, so fixed 1e-3 bounds on early-layer gradients
    would sit below the reference's own fp32 noise floor."""
    out = {}
    torch.manual_seed(model_seed)
    m = factory(**kwargs)
    m.train()
    is_conv_net = x.dim() == 4 and any(isinstance(mod, torch.nn.BatchNorm2d) for mod in m.modules())
    if is_conv_net:
        m = m.to(memory_format=torch.channels_last)
        lg = m(x.contiguous(memory_format=torch.channels_last))
        criterion(lg, y).backward()
        worst = 0.0
        for n, p in m.named_parameters():
            if float(p.grad.norm()) > 1e-7:
                worst = max(worst, _rel(p.grad.flatten()[:64], samples[n]))
                if n in full:
                    worst = max(worst, _rel(p.grad, full[n]))
        out['fp32_reorder_logits'] = _rel(lg.detach(), logits)
        out['fp32_reorder_grad_sample'] = worst
    torch.manual_seed(model_seed)
    m = factory(**kwargs)
    m.train()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        lg = m(x)
        ls = criterion(lg, y)
    ls.backward()
    a = torch.cat([p.grad.flatten()[:64].double() for n, p in m.named_parameters()])
    b = torch.cat([samples[n].double() for n, p in m.named_parameters()])
    out['bf16_logits'] = _rel(lg.detach().float(), logits)
    out['bf16_loss'] = abs(float(ls) - loss) / abs(loss)
    out['bf16_grad_sample_cos'] = float(a @ b / (a.norm() * b.norm()))
    return out


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
    noise = measure_reference_noise(factory, kwargs, x, y, criterion, model_seed, samples, full, logits.detach(), float(loss))
    fx = {
        'name': name, 'kwargs': kwargs, 'shape': list(shape), 'num_classes': num_classes, 'soft': soft,
        'model_seed': model_seed, 'data_seed': data_seed,
        'input_checksum': float(x.double().sum()), 'label_checksum': float(y.double().sum()),
        'logits': logits.detach().clone(), 'loss': float(loss), 'eval_logits': eval_logits.clone(),
        'grad_norm': norms, 'grad_sample': samples, 'grad_full': full, 'buffers_after': buffers,
        'reference_noise': noise,
        'torch_version': torch.__version__,
    }
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.pt')
    torch.save(fx, path)
    print(f'{name}: loss={float(loss):.6f} logits_norm={float(logits.norm()):.5f} -> {path} '
          f'({os.path.getsize(path) / 1024:.0f} KiB) noise={ {k: round(v, 5) for k, v in noise.items()} }')


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    from SimpleAICV.classification import backbones, losses   # the reference's own modules

    ce, soft_ce = losses.CELoss(), losses.OneHotLabelCELoss()
    run_case('resnet18cifar_b8', backbones.resnet18cifar, {'num_classes': 100}, (8, 3, 32, 32), 100, ce, False)
    run_case('resnet18cifar_b64', backbones.resnet18cifar, {'num_classes': 100}, (64, 3, 32, 32), 100, ce, False)
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
