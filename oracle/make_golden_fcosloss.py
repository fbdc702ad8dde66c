"""Generates tests/golden/fcos_loss.pt by RUNNING THE REFERENCE FCOSLoss (SimpleAICV/detection/losses.py:434-842, imported from
/root/reference) on the CPU in fp32: seeded head outputs of a 128 x 160 image pyramid (five levels, 8 classes), three images with
6 / 0 / 14 ground-truth boxes padded to 16 rows -- the three loss values, gradient norms + samples for every head tensor, and the
per-point class targets / centre-ness of get_batch_position_annotations, for GIoU with centre sampling (the default), CIoU, and
IoU without centre sampling.  Inputs come from torch.rand only (bit-reproducible); the test rebuilds them with `inputs()`.

Build container only:   python oracle/make_golden_fcosloss.py"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
SIZES = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
CLASSES, BATCH, ROWS = 8, 3, 16
# regression ranges scaled to the small image so that several levels own boxes
MI = [[-1, 24], [24, 48], [48, 96], [96, 192], [192, 100000000]]
CASES = {'giou': dict(mi=MI), 'ciou': dict(mi=MI, box_loss_iou_type='CIoU', gamma=1.5, alpha=0.3, cls_loss_weight=2.0, center_ness_loss_weight=0.5),
         'iou_nocenter': dict(mi=MI, box_loss_iou_type='IoU', use_center_sample=False), 'default_ranges': dict(),
         'default_ranges_nocenter': dict(use_center_sample=False, box_loss_iou_type='EIoU')}


def inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    cls, reg, ctr = [], [], []
    for h, w in SIZES:
        p = torch.rand(BATCH, h, w, CLASSES, generator=g) * 0.998 + 0.001
        p[:, :, :, 0] = torch.where(torch.rand(BATCH, h, w, generator=g) < 0.3, torch.full((BATCH, h, w), 5e-5), p[:, :, :, 0])
        cls.append(p)
        reg.append(torch.rand(BATCH, h, w, 4, generator=g) * 3.0 + 1.0)          # log-distances: exp() = 2.7 .. 55 pixels
        ctr.append(torch.rand(BATCH, h, w, 1, generator=g) * 0.998 + 0.001)
    annots = -torch.ones(BATCH, ROWS, 5)
    for b, n in enumerate([6, 0, 14]):
        cx = torch.rand(n, generator=g) * 120 + 20
        cy = torch.rand(n, generator=g) * 90 + 19
        bw = torch.rand(n, generator=g) * 100 + 16
        bh = torch.rand(n, generator=g) * 80 + 16
        annots[b, :n, 0] = (cx - bw / 2).clamp(min=0)
        annots[b, :n, 1] = (cy - bh / 2).clamp(min=0)
        annots[b, :n, 2] = (cx + bw / 2).clamp(max=159)
        annots[b, :n, 3] = (cy + bh / 2).clamp(max=127)
        annots[b, :n, 4] = torch.randint(0, CLASSES, (n,), generator=g).float()
    return cls, reg, ctr, annots


def sample_idx(numel, k=24):
    return torch.linspace(0, numel - 1, min(k, numel)).long()


def main():
    sys.path.insert(0, REF)
    for name in ['cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval',
                 'pycocotools.coco', 'calflops']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from SimpleAICV.detection.losses import FCOSLoss
    out = {}
    for name, kw in CASES.items():
        cls, reg, ctr, annots = inputs()
        leaves = [t.requires_grad_(True) for t in cls + reg + ctr]
        crit = FCOSLoss(**kw)
        losses = crit([cls, reg, ctr], annots)
        sum(losses.values()).backward()
        sizes = [[t.shape[2], t.shape[1]] for t in cls]
        pos = [torch.tensor(p).unsqueeze(0).repeat(BATCH, 1, 1, 1) for p in crit.positions(sizes)]
        with torch.no_grad():
            _, _, _, tg = crit.get_batch_position_annotations([t.detach() for t in cls], [t.detach() for t in reg], [t.detach() for t in ctr],
                                                              pos, annots, use_center_sample=crit.use_center_sample)
        out[name] = {'config': kw, 'losses': {k: float(v) for k, v in losses.items()},
                     'grad_norm': [float(t.grad.norm()) for t in leaves],
                     'grad_sample': [t.grad.flatten()[sample_idx(t.numel())].clone() for t in leaves],
                     'class_targets': tg[:, :, 4].to(torch.int8).clone(), 'centerness': tg[:, :, 5].clone(), 'ltrb': tg[:, :, 0:4].clone()}
        print(name, out[name]['losses'], 'positives', int((tg[:, :, 4] > 0).sum()), [int((tg[b, :, 4] > 0).sum()) for b in range(BATCH)])
    torch.save(out, os.path.join(OUT, 'fcos_loss.pt'))
    print('bytes', os.path.getsize(os.path.join(OUT, 'fcos_loss.pt')))


if __name__ == '__main__':
    main()
