"""Synthetic stand-in for the reference's CocoDetection dataset + transform block in the benchmark configs: a sample has
the contract the reference hands to DETRDetectionCollater AFTER its transforms (detection/common.py:291-363):
{'image': float32 HWC (normalised), 'annots': float32 [n, 5] xyxy + class in pixels, 'scale': float, 'size': [h, w]}."""
import numpy as np
from torch.utils.data import Dataset


class SyntheticDetectionDataset(Dataset):

    def __init__(self, num_samples, height, width, num_classes=80, max_boxes=20, seed=0):
        self.num_samples, self.height, self.width = num_samples, height, width
        self.num_classes, self.max_boxes, self.seed = num_classes, max_boxes, seed

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx):
        rng = np.random.default_rng((self.seed, idx))
        h, w = self.height, self.width
        image = rng.standard_normal((h, w, 3), dtype=np.float32)
        n = int(rng.integers(1, self.max_boxes + 1))
        cx, cy = rng.uniform(0.25, 0.75, n) * w, rng.uniform(0.25, 0.75, n) * h
        bw, bh = rng.uniform(0.05, 0.5, n) * w, rng.uniform(0.05, 0.5, n) * h
        boxes = np.stack([np.clip(cx - bw / 2, 0, w - 1), np.clip(cy - bh / 2, 0, h - 1), np.clip(cx + bw / 2, 1, w),
                          np.clip(cy + bh / 2, 1, h), rng.integers(0, self.num_classes, n)], axis=1).astype(np.float32)
        return {'image': image, 'annots': boxes, 'scale': np.float32(1.0), 'size': np.array([h, w], dtype=np.float32)}
