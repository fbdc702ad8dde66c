"""Fixture for the Mixup / CutMix collater, produced by the reference collater itself
(SimpleAICV/classification/mixupcutmixclassificationcollator.py:99-284) under fixed numpy seeds.
    python oracle/make_golden_mixup.py  ->  tests/golden/mixup_cutmix.pt      (build container only)"""
import os
import sys

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'mixup_cutmix.pt')

CASES = [dict(mode=m, **kw) for m in ('batch', 'pair', 'elem')
         for kw in (dict(), dict(cutmix_alpha=0.), dict(mixup_alpha=0.), dict(cutmix_minmax=[0.2, 0.8]),
                    dict(mixup_cutmix_prob=0.5), dict(correct_lam=False, label_smoothing=0.0))]


def batch(seed, n=6, h=12, w=10, classes=10):
    r = np.random.default_rng(seed)
    return [{'image': r.standard_normal((h, w, 3)).astype(np.float32), 'label': int(r.integers(0, classes))} for _ in range(n)]


def main():
    sys.path.insert(0, REF)
    from SimpleAICV.classification.mixupcutmixclassificationcollator import MixupCutmixClassificationCollater
    out = []
    for ci, kw in enumerate(CASES):
        for seed in (0, 1):
            np.random.seed(100 * ci + seed)
            r = MixupCutmixClassificationCollater(num_classes=10, **kw)(batch(seed))
            out.append({'kwargs': kw, 'np_seed': 100 * ci + seed, 'data_seed': seed, 'image': r['image'].contiguous(),
                        'label': r['label']})
    torch.save(out, OUT)
    print('wrote', OUT, len(out), 'cases', os.path.getsize(OUT) // 1024, 'KiB')


if __name__ == '__main__':
    main()
