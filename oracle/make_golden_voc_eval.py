"""Generates tests/golden/voc_eval.pt by RUNNING THE REFERENCE evaluate_voc_detection (tools/scripts.py:559-739, imported from
/root/reference) on the CPU with a stub model / criterion / decoder that replay seeded detections: 12 images in 3 batches, 6 classes,
ground truth with 0..7 boxes per image, detections that are jittered copies of ground-truth boxes (some duplicated: only the first
may match), pure false positives, images without detections; a second variant has a class without ground truth (the reference's
recall is 0 / 0 there and every mAP becomes NaN -- kept as is).  The fixture holds the result dict
(mAP per threshold, per-class AP); the test replays the same stubs through this repository's evaluate_voc_detection.

Build container only:   python oracle/make_golden_voc_eval.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
CLASSES, IMAGES, BATCH, MAX_DET, ROWS = 6, 12, 4, 20, 8
THRESHOLDS = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]


def dataset(seed=0, gtless_class=False):
    """per batch: {'image', 'annots', 'scale', 'size'} and the detections the stub decoder returns for it"""
    rng = np.random.RandomState(seed)
    batches = []
    for b in range(IMAGES // BATCH):
        annots = -np.ones((BATCH, ROWS, 5), dtype=np.float32)
        scores = -np.ones((BATCH, MAX_DET), dtype=np.float32)
        classes = -np.ones((BATCH, MAX_DET), dtype=np.float32)
        boxes = np.zeros((BATCH, MAX_DET, 4), dtype=np.float32)
        scales = (rng.rand(BATCH) * 0.5 + 0.75).astype(np.float32)
        sizes = np.stack([rng.randint(200, 300, BATCH), rng.randint(220, 340, BATCH)], axis=1).astype(np.float32)     # (h, w) original
        for i in range(BATCH):
            n = rng.randint(0, ROWS)
            h, w = sizes[i] * scales[i]
            for k in range(n):
                x1, y1 = rng.rand() * (w - 60), rng.rand() * (h - 60)
                annots[i, k] = [x1, y1, x1 + 20 + rng.rand() * 40, y1 + 20 + rng.rand() * 40, rng.randint(0, CLASSES - 1 if gtless_class else CLASSES)]   # gtless_class: the last class never has ground truth
            det = []
            for k in range(n):
                if rng.rand() < 0.8:
                    jit = rng.randn(4) * rng.choice([1.0, 4.0, 9.0])
                    det.append((annots[i, k, :4] + jit, annots[i, k, 4] if rng.rand() < 0.9 else rng.randint(0, CLASSES), rng.rand()))
                    if rng.rand() < 0.3:
                        det.append((annots[i, k, :4] + rng.randn(4), annots[i, k, 4], rng.rand()))
            for _ in range(rng.randint(0, 4)):
                x1, y1 = rng.rand() * (w - 50), rng.rand() * (h - 50)
                det.append((np.array([x1, y1, x1 + 30, y1 + 30]), rng.randint(0, CLASSES), rng.rand()))
            if b == 1 and i == 2:
                det = []
            det.sort(key=lambda d: -d[2])
            for k, (bx, c, s) in enumerate(det[:MAX_DET]):
                boxes[i, k], classes[i, k], scores[i, k] = bx, c, s
        batches.append(({'image': torch.zeros(BATCH, 3, 8, 8), 'annots': torch.from_numpy(annots), 'scale': scales, 'size': sizes},
                        (scores, classes, boxes)))
    return batches


class StubModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.calls = 0

    def forward(self, images):
        self.calls += 1
        return self.calls - 1                         # the batch index: the stub decoder looks its detections up


class StubConfig:
    network = 'resnet50_retinanet'
    num_classes = CLASSES
    batch_size = BATCH
    gpus_num = 1
    eval_type = 'VOC'
    eval_voc_iou_threshold_list = THRESHOLDS


def stubs(gtless_class=False):
    batches = dataset(gtless_class=gtless_class)
    model = StubModel()
    criterion = lambda outs, annots: {'cls_loss': torch.tensor(0.5 + 0.25 * outs), 'reg_loss': torch.tensor(0.125)}
    decoder = lambda outs: tuple(a.copy() for a in batches[outs][1])
    return [b[0] for b in batches], model, criterion, decoder, StubConfig()


def main():
    sys.path.insert(0, REF)
    for name in ['cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval',
                 'pycocotools.coco', 'calflops', 'thop']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['pycocotools.cocoeval'].COCOeval = object
    sys.modules['pycocotools'].mask = sys.modules['pycocotools.mask']
    torch.cuda.synchronize = lambda *a, **k: None
    import tools.scripts as S
    S.tqdm = lambda it, *a, **k: it
    out = {}
    for name, gtless in (('all_classes', False), ('class_without_ground_truth', True)):
        loader, model, criterion, decoder, config = stubs(gtless)
        res = S.evaluate_voc_detection(loader, model, criterion, decoder, config)
        out[name] = {k: (float(v) if not isinstance(v, (str, dict)) else (dict((int(c), float(a)) for c, a in v.items()) if isinstance(v, dict) else v))
                     for k, v in res.items()}
        print(name, {k[:8]: round(v, 3) for k, v in out[name].items() if k.endswith('mAP')}, 'test_loss', out[name]['test_loss'])
    torch.save(out, os.path.join(OUT, 'voc_eval.pt'))


if __name__ == '__main__':
    main()
