"""Generates tests/golden/fcos_r18_tiny.pt and anchors_positions.pt by RUNNING THE REFERENCE (imported from /root/reference):
resnet18_fcos (20 classes) on a seeded batch of 2 x 3 x 128 x 160 in fp32 -- the fifteen outputs of
SimpleAICV/detection/models/fcos.py:52-90, a scalar of them back-propagated (per-parameter gradient norms + samples); and the
host-side tables of models/anchor.py (RetinaAnchors / FCOSPositions) for a square and a ragged image as SHA-256 digests of the
float32 bytes plus corner samples.

Build container only:   python oracle/make_golden_fcos.py"""
import hashlib
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
BATCH, H, W = 2, 128, 160
STRIDES = [8, 16, 32, 64, 128]
IMAGES = [(640, 640), (600, 433)]            # (w, h)


def sample_idx(numel, k=16):
    return torch.linspace(0, numel - 1, min(k, numel)).long()


def scalar_of(groups, g):
    s = 0.
    for heads in groups:
        for t in heads:
            s = s + (t.float() * torch.randn(t.shape, generator=g)).sum() / t.numel() ** 0.5
    return s


def digest(a):
    a = np.ascontiguousarray(a)
    return {'shape': tuple(a.shape), 'dtype': str(a.dtype), 'sha256': hashlib.sha256(a.tobytes()).hexdigest(),
            'first': torch.from_numpy(a.reshape(-1)[:8].copy()), 'last': torch.from_numpy(a.reshape(-1)[-8:].copy())}


def main():
    sys.path.insert(0, REF)
    for name in ['cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval',
                 'pycocotools.coco', 'calflops']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from SimpleAICV.detection.models import fcos
    from SimpleAICV.detection.models.anchor import RetinaAnchors, FCOSPositions
    torch.manual_seed(0)
    model = fcos.resnet18_fcos(num_classes=20)
    with torch.no_grad():
        model.scales.copy_(torch.tensor([0.9, 1.0, 1.1, 1.2, 0.8]))       # distinct per-level scales (set after construction on both sides)
    model.train()
    init = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randn(BATCH, H, W, 3, generator=torch.Generator().manual_seed(1)).permute(0, 3, 1, 2)
    outs = model(x)
    loss = scalar_of(outs, torch.Generator().manual_seed(2))
    loss.backward()
    fx = {'config': dict(num_classes=20), 'input_shape': (BATCH, 3, H, W), 'outs': [[t.detach() for t in heads] for heads in outs],
          'scalar': float(loss.detach()),
          'init_sample': {k: v.flatten()[sample_idx(v.numel())].clone() for k, v in init.items() if v.dtype.is_floating_point},
          'grad_norm': {k: float(p.grad.norm()) for k, p in model.named_parameters() if p.grad is not None},
          'grad_sample': {k: p.grad.flatten()[sample_idx(p.numel())].clone() for k, p in model.named_parameters() if p.grad is not None}}
    torch.save(fx, os.path.join(OUT, 'fcos_r18_tiny.pt'))
    print('fcos', [[tuple(t.shape) for t in heads] for heads in outs][0], float(loss), len(fx['grad_norm']))

    tables = {}
    for (w, h) in IMAGES:
        sizes = [[math.ceil(w / s), math.ceil(h / s)] for s in STRIDES]
        tables[(w, h)] = {'sizes': sizes, 'anchors': [digest(a) for a in RetinaAnchors()(sizes)],
                          'positions': [digest(a) for a in FCOSPositions()(sizes)]}
    custom = RetinaAnchors(areas=[[24, 24], [48, 48], [96, 96]], ratios=[0.4, 1.6], scales=[1.0, 1.5], strides=[8, 16, 32])
    sizes = [[13, 9], [7, 5], [4, 3]]
    tables['custom'] = {'sizes': sizes, 'anchors': [digest(a) for a in custom(sizes)]}
    torch.save(tables, os.path.join(OUT, 'anchors_positions.pt'))
    print('tables', list(tables))


if __name__ == '__main__':
    main()
