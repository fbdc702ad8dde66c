"""Generates tests/golden/retina_loss.pt by RUNNING THE REFERENCE RetinaLoss (SimpleAICV/detection/losses.py:123-433, imported from
/root/reference) on the CPU in fp32: seeded head outputs of a 128 x 160 image pyramid (five levels, 9 anchors, 8 classes), three
images with 6 / 0 / 14 ground-truth boxes padded to 16 rows -- loss values, gradient norms + samples for every head tensor, and the
class-target census of get_batch_anchors_annotations (exact integers), for box_loss_type SmoothL1 / GIoU / CIoU and two focal
settings.  Inputs are built from torch.rand only (bit-reproducible on every CPU); the test rebuilds them with `inputs()` below.

Build container only:   python oracle/make_golden_retinaloss.py"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
SIZES = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]          # (h, w) of a 128 x 160 image at strides 8 .. 128
CLASSES, ANCHORS, BATCH, ROWS = 8, 9, 3, 16
CASES = {'smoothl1': dict(box_loss_type='SmoothL1'), 'giou': dict(box_loss_type='GIoU'), 'ciou': dict(box_loss_type='CIoU'),
         'smoothl1_gamma15': dict(box_loss_type='SmoothL1', gamma=1.5, alpha=0.3, beta=0.2, cls_loss_weight=2.0, box_loss_weight=0.5)}


def inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    cls, reg = [], []
    for h, w in SIZES:
        p = torch.rand(BATCH, h, w, ANCHORS, CLASSES, generator=g) * 0.998 + 0.001
        p[:, :, :, 0, 0] = 5e-5                    # below / above the clamp range: zero gradient there
        p[:, :, :, 1, 1] = 0.99995
        cls.append(p)
        reg.append(torch.rand(BATCH, h, w, ANCHORS, 4, generator=g) * 0.6 - 0.3)
    annots = -torch.ones(BATCH, ROWS, 5)
    for b, n in enumerate([6, 0, 14]):
        cx = torch.rand(n, generator=g) * 120 + 20
        cy = torch.rand(n, generator=g) * 90 + 19
        bw = torch.rand(n, generator=g) * 90 + 24
        bh = torch.rand(n, generator=g) * 70 + 24
        annots[b, :n, 0] = (cx - bw / 2).clamp(min=0)
        annots[b, :n, 1] = (cy - bh / 2).clamp(min=0)
        annots[b, :n, 2] = (cx + bw / 2).clamp(max=159)
        annots[b, :n, 3] = (cy + bh / 2).clamp(max=127)
        annots[b, :n, 4] = torch.randint(0, CLASSES, (n,), generator=g).float()
    return cls, reg, annots


def sample_idx(numel, k=24):
    return torch.linspace(0, numel - 1, min(k, numel)).long()


def main():
    sys.path.insert(0, REF)
    for name in ['cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval', 'pycocotools.coco', 'calflops']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from SimpleAICV.detection.losses import RetinaLoss
    out = {}
    for name, kw in CASES.items():
        cls, reg, annots = inputs()
        leaves = [t.requires_grad_(True) for t in cls + reg]
        crit = RetinaLoss(**kw)
        losses = crit([cls, reg], annots)
        (losses['cls_loss'] + losses['reg_loss']).backward()
        # census of the assignment (the reference's own method on the same anchors)
        sizes = [[t.shape[2], t.shape[1]] for t in cls]
        table = torch.cat([torch.tensor(a).view(-1, 4) for a in crit.anchors(sizes)], dim=0)
        tg = crit.get_batch_anchors_annotations(table.unsqueeze(0).repeat(BATCH, 1, 1), annots)
        out[name] = {'config': kw, 'cls_loss': float(losses['cls_loss']), 'reg_loss': float(losses['reg_loss']),
                     'grad_norm': [float(t.grad.norm()) for t in leaves],
                     'grad_sample': [t.grad.flatten()[sample_idx(t.numel())].clone() for t in leaves],
                     'census': [[int((tg[b, :, 4] == -1).sum()), int((tg[b, :, 4] == 0).sum()), int((tg[b, :, 4] > 0).sum())] for b in range(BATCH)],
                     'class_targets': tg[:, :, 4].to(torch.int8).clone(),
                     'box_targets_sample': tg[0, :, 0:4][sample_idx(tg.shape[1], 64)].clone()}
        print(name, out[name]['cls_loss'], out[name]['reg_loss'], out[name]['census'])
    torch.save(out, os.path.join(OUT, 'retina_loss.pt'))
    print('bytes', os.path.getsize(os.path.join(OUT, 'retina_loss.pt')))


if __name__ == '__main__':
    main()
