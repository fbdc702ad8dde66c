"""Generates tests/golden/detection_collater.pt by RUNNING THE REFERENCE DetectionCollater (SimpleAICV/detection/common.py:243-288,
imported from /root/reference) on three seeded samples (ragged image sizes, 0 / 2 / 5 annotation rows) for the yolo-style and
retina-style canvases.  Build container only:   python oracle/make_golden_collater.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
CASES = {'yolo': dict(resize=32, resize_type='yolo_style', max_annots_num=6), 'retina': dict(resize=24, resize_type='retina_style', max_annots_num=8)}


def samples(seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for (h, w), n in zip([(20, 32), (32, 17), (9, 9)], [0, 2, 5]):
        a = np.zeros((n, 5), dtype=np.float32)
        if n:
            a[:, 0:2] = rng.rand(n, 2) * 5
            a[:, 2:4] = a[:, 0:2] + 1 + rng.rand(n, 2) * 3
            a[:, 4] = rng.randint(0, 20, n)
        out.append({'image': rng.rand(h, w, 3).astype(np.float32), 'annots': a, 'scale': np.float32(0.5 + rng.rand()),
                    'size': np.array([h * 2, w * 2], dtype=np.float32)})
    return out


def main():
    sys.path.insert(0, REF)
    for name in ['cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval',
                 'pycocotools.coco', 'calflops']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from SimpleAICV.detection.common import DetectionCollater
    out = {}
    for name, kw in CASES.items():
        r = DetectionCollater(**kw)(samples())
        out[name] = {'config': kw, 'image': r['image'].clone(), 'image_stride': tuple(r['image'].stride()), 'annots': r['annots'].clone(),
                     'scale': torch.from_numpy(r['scale']), 'size': torch.from_numpy(r['size'])}
        print(name, tuple(r['image'].shape), r['image'].stride())
    torch.save(out, os.path.join(OUT, 'detection_collater.pt'))


if __name__ == '__main__':
    main()
