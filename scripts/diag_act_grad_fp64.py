"""Gradient w.r.t. every ConvBnActBlock output / residual-block output of resnet18cifar (batch 8): HIP fp32 vs the float64 oracle
(and the fp32 oracle), relative L2 -- where along the backward pass the HIP path leaves the float64 trajectory."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones, losses  # noqa: E402
from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import ConvBnActBlock  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

name, batch, size = 'resnet18cifar', 8, 32
torch.manual_seed(0)
model = backbones.__dict__[name](num_classes=100)
sd = {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in model.state_dict().items()}
pnames = [n for n, _ in model.named_parameters()]
g = torch.Generator().manual_seed(1)
x = torch.randn(batch, 3, size, size, generator=g)
y = torch.randint(0, 100, (batch,), generator=g)


def oracle_act_grads(dtype):
    kept = {}
    orig_cba, orig_bb = O.conv_bn_act, O.basic_block

    def cba(xx, s, prefix, *a, **k):
        out = orig_cba(xx, s, prefix, *a, **k)
        if prefix.endswith('conv1'):
            out.retain_grad()
            kept[prefix] = out
        return out

    def bb(xx, s, prefix, *a, **k):
        out = orig_bb(xx, s, prefix, *a, **k)
        out.retain_grad()
        kept[prefix + '.conv2'] = out           # the HIP conv2 block returns the residual block's output
        return out

    O.conv_bn_act, O.basic_block = cba, bb
    try:
        s2 = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()}
        O.loss_and_grads(lambda leaves, inp: O.resnet_forward(name, leaves, inp, training=True), s2, pnames, x.to(dtype),
                         loss_fn=O.ce_loss, label=y)
    finally:
        O.conv_bn_act, O.basic_block = orig_cba, orig_bb
    return {k: v.grad.detach().double() for k, v in kept.items()}


a64, a32 = oracle_act_grads(torch.float64), oracle_act_grads(torch.float32)
model = model.cuda()
got = {}
for n, m in model.named_modules():
    if isinstance(m, ConvBnActBlock):
        def hook(mod, inp, out, n=n):
            t = out[0] if isinstance(out, tuple) else out
            if t.requires_grad:
                t.register_hook(lambda gr, n=n: got.__setitem__(n, gr.detach().double().cpu()))
        m.register_forward_hook(hook)
losses.CELoss()(model(x.cuda()), y.cuda()).backward()
torch.cuda.synchronize()
print(f'{"d loss / d output of":36s} {"|g|":>10s} {"HIP vs f64":>11s} {"cpu32 vs f64":>12s} {"mean err / |g| rms":>18s}')
for k in a64:
    if k not in got:
        continue
    r, h = a64[k], got[k].contiguous()
    eh = float((h - r).norm() / r.norm())
    ec = float((a32[k] - r).norm() / r.norm())
    # is the error a per-channel constant? mean over (N, H, W) of the error, relative to the rms of the gradient
    off = float((h - r).mean(dim=(0, 2, 3)).abs().max() / r.pow(2).mean().sqrt())
    print(f'{k:36s} {float(r.norm()):10.3e} {eh:11.2e} {ec:12.2e} {off:18.2e}')

# where exactly: the deepest output (first in backward order) whose gradient is off, element by element
order = list(a64.keys())[::-1]
for k in order:
    if k not in got:
        continue
    r, h = a64[k], got[k].contiguous()
    if float((h - r).norm() / r.norm()) > 1e-4:
        d = (h - r).abs()
        rms = float(r.pow(2).mean().sqrt())
        bad = (d > 1e-2 * rms).nonzero()
        print(f'\nfirst bad output in backward order: {k}, shape {tuple(r.shape)}, rms {rms:.3e}; {bad.shape[0]} elements off by > 1e-2 rms '
              f'(of {r.numel()}); share of the squared error they carry: {float(d[d > 1e-2 * rms].pow(2).sum() / d.pow(2).sum()):.3f}')
        import collections
        print('  by image:', dict(collections.Counter(bad[:, 0].tolist())))
        print('  by channel (top 8):', collections.Counter(bad[:, 1].tolist()).most_common(8))
        print('  by row h (top 8):', collections.Counter(bad[:, 2].tolist()).most_common(8))
        print('  by col w (top 8):', collections.Counter(bad[:, 3].tolist()).most_common(8))
        for idx in bad[:12].tolist():
            n_, c_, h_, w_ = idx
            print(f'   [{n_},{c_},{h_},{w_}] hip {float(h[n_, c_, h_, w_]):+.6e} f64 {float(r[n_, c_, h_, w_]):+.6e}')
        break
