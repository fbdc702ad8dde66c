"""Generates tests/golden/dense_decoders.pt by RUNNING THE REFERENCE decoders (SimpleAICV/detection/decode.py RetinaDecoder :174-270,
FCOSDecoder :273-363, DecodeMethod / DetNMSMethod :25-171; imported from /root/reference) on seeded head outputs of a 128 x 160 image
pyramid (batch 3): python_nms, diou_python_nms, a low top-n and a high threshold.  Inputs come from torch.rand only
(bit-reproducible); a few anchors per image get high scores and sensible offsets so that NMS has clusters to resolve.

Build container only:   python oracle/make_golden_decoders.py"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
SIZES = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
CLASSES, ANCHORS, BATCH = 8, 9, 3
RETINA = {'python': dict(), 'diou': dict(nms_type='diou_python_nms', nms_threshold=0.4), 'top50': dict(topn=50, max_object_num=20),
          'thr30': dict(min_score_threshold=0.3)}
FCOS = {'python': dict(), 'diou': dict(nms_type='diou_python_nms', nms_threshold=0.5), 'top50': dict(topn=50, max_object_num=20)}


def retina_inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    cls, reg = [], []
    for h, w in SIZES:
        p = torch.rand(BATCH, h, w, ANCHORS, CLASSES, generator=g) ** 4            # mostly small scores, a tail of confident ones
        cls.append(p)
        reg.append(torch.rand(BATCH, h, w, ANCHORS, 4, generator=g) * 0.8 - 0.4)
    return cls, reg


def fcos_inputs(seed=1):
    g = torch.Generator().manual_seed(seed)
    cls, reg, ctr = [], [], []
    for h, w in SIZES:
        cls.append(torch.rand(BATCH, h, w, CLASSES, generator=g) ** 3)
        reg.append(torch.rand(BATCH, h, w, 4, generator=g) * 2.5 + 1.5)
        ctr.append(torch.rand(BATCH, h, w, 1, generator=g))
    return cls, reg, ctr


def main():
    sys.path.insert(0, REF)
    for name in ['cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval',
                 'pycocotools.coco', 'calflops']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision.ops'].nms = None
    from SimpleAICV.detection.decode import RetinaDecoder, FCOSDecoder
    out = {'retina': {}, 'fcos': {}}
    for name, kw in RETINA.items():
        s, c, b = RetinaDecoder(**kw)(retina_inputs())
        out['retina'][name] = {'config': kw, 'scores': torch.from_numpy(s), 'classes': torch.from_numpy(c), 'boxes': torch.from_numpy(b)}
        print('retina', name, [(int((s[i] >= 0).sum())) for i in range(BATCH)])
    for name, kw in FCOS.items():
        s, c, b = FCOSDecoder(**kw)(fcos_inputs())
        out['fcos'][name] = {'config': kw, 'scores': torch.from_numpy(s), 'classes': torch.from_numpy(c), 'boxes': torch.from_numpy(b)}
        print('fcos', name, [(int((s[i] >= 0).sum())) for i in range(BATCH)])
    torch.save(out, os.path.join(OUT, 'dense_decoders.pt'))
    print('bytes', os.path.getsize(os.path.join(OUT, 'dense_decoders.pt')))


if __name__ == '__main__':
    main()
