"""Generates tests/golden/retinanet_r18_tiny.pt and fcos_head_tiny.pt by RUNNING THE REFERENCE (imported from /root/reference):
resnet18_retinanet (20 classes) on a seeded batch of 2 x 3 x 128 x 160 in fp32 -- the ten outputs of
SimpleAICV/detection/models/retinanet.py:60-97, a scalar of them back-propagated (per-parameter gradient norms + samples), the BN
buffers; and FCOSClsRegCntHead (head.py:88-181, GroupNorm towers) forward / backward on one pyramid level.

Build container only:   python oracle/make_golden_retinanet.py
The reference detection package imports cv2 / torchvision at module scope for dataset code; empty stand-ins are registered first
(as in make_golden_detr.py).  The product-side test builds the same models under the same seed (identical initial weights)."""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')

RETINA = dict(num_classes=20)
BATCH, H, W = 2, 128, 160


def scalar_of(cls_heads, reg_heads, g):
    """A fixed random projection of every output (weights from generator g): one scalar whose gradient reaches every parameter."""
    s = 0.
    for t in list(cls_heads) + list(reg_heads):
        s = s + (t.float() * torch.randn(t.shape, generator=g)).sum() / t.numel() ** 0.5
    return s


def sample_idx(numel, k=16):
    return torch.linspace(0, numel - 1, min(k, numel)).long()


def main():
    sys.path.insert(0, REF)
    for name in ['cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval',
                 'pycocotools.coco', 'calflops']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from SimpleAICV.detection.models import retinanet
    from SimpleAICV.detection.models.head import FCOSClsRegCntHead
    torch.manual_seed(0)
    model = retinanet.resnet18_retinanet(**RETINA)
    model.train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(BATCH, H, W, 3, generator=g).permute(0, 3, 1, 2)          # NHWC memory, as the collaters deliver
    init = {k: v.clone() for k, v in model.state_dict().items()}
    cls_heads, reg_heads = model(x)
    gw = torch.Generator().manual_seed(2)
    loss = scalar_of(cls_heads, reg_heads, gw)
    loss.backward()
    fx = {'config': RETINA, 'input_shape': (BATCH, 3, H, W), 'cls': [t.detach() for t in cls_heads], 'reg': [t.detach() for t in reg_heads],
          'scalar': float(loss), 'init_sample': {k: v.flatten()[sample_idx(v.numel())].clone() for k, v in init.items() if v.dtype.is_floating_point},
          'grad_norm': {k: float(p.grad.norm()) for k, p in model.named_parameters() if p.grad is not None},
          'grad_sample': {k: p.grad.flatten()[sample_idx(p.numel())].clone() for k, p in model.named_parameters() if p.grad is not None},
          'bn_buffers': {k: v.clone() for k, v in model.state_dict().items() if 'running_' in k and 'layer4.1' in k}}
    # how far the reference's OWN bf16 autocast run moves from its fp32 run (the gate of the product's bf16 test)
    torch.manual_seed(0)
    model2 = retinanet.resnet18_retinanet(**RETINA)
    model2.train()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        c2, r2 = model2(x)
    def rel(a, b):
        return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))
    fx['bf16_dev'] = {'cls': [rel(a, b) for a, b in zip(c2, fx['cls'])], 'reg': [rel(a, b) for a, b in zip(r2, fx['reg'])]}
    print('reference bf16 deviation', fx['bf16_dev'])
    torch.save(fx, os.path.join(OUT, 'retinanet_r18_tiny.pt'))
    print('retinanet', [tuple(t.shape) for t in cls_heads], float(loss), len(fx['grad_norm']))

    torch.manual_seed(3)
    head = FCOSClsRegCntHead(64, 20, num_layers=2, use_gn=True, cnt_on_reg=True)
    g = torch.Generator().manual_seed(4)
    f = torch.randn(2, 24, 20, 64, generator=g).permute(0, 3, 1, 2).requires_grad_(True)
    outs = head(f)
    gw = torch.Generator().manual_seed(5)
    s = sum((t.float() * torch.randn(t.shape, generator=gw)).sum() for t in outs)
    s.backward()
    torch.save({'outs': [t.detach() for t in outs], 'dx': f.grad.clone(),
                'grad_norm': {k: float(p.grad.norm()) for k, p in head.named_parameters()},
                'init_sample': {k: v.flatten()[sample_idx(v.numel())].clone() for k, v in head.state_dict().items()}},
               os.path.join(OUT, 'fcos_head_tiny.pt'))
    print('fcos head', [tuple(t.shape) for t in outs])


if __name__ == '__main__':
    main()
