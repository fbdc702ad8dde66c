"""Synthetic stand-in for the reference's SAMSegmentationDataset + transform block in the benchmark config: a sample has the
contract handed to SAMBatchCollater after the transforms (interactive_segmentation/common.py:129-232): image float32 HWC,
a binary ellipse mask, its box, one positive prompt point inside it, a noisy prompt box and the mask as prompt mask."""
import numpy as np
from torch.utils.data import Dataset


class SyntheticSAMDataset(Dataset):

    def __init__(self, num_samples, image_size=1024, seed=0):
        self.num_samples, self.image_size, self.seed = num_samples, image_size, seed

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx):
        rng = np.random.default_rng((self.seed, idx))
        s = self.image_size
        image = rng.standard_normal((s, s, 3), dtype=np.float32)
        cy, cx = rng.uniform(0.3, 0.7, 2) * s
        ry, rx = rng.uniform(0.08, 0.25, 2) * s
        yy, xx = np.mgrid[0:s, 0:s]
        mask = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1.0).astype(np.float32)
        box = np.array([cx - rx, cy - ry, cx + rx, cy + ry], dtype=np.float32)
        noise = rng.uniform(-0.1, 0.1, 4).astype(np.float32) * np.array([2 * rx, 2 * ry, 2 * rx, 2 * ry], dtype=np.float32)
        return {'image': image, 'box': box, 'mask': mask, 'size': np.array([s, s], dtype=np.float32),
                'prompt_point': np.array([[cx, cy, 1.0]], dtype=np.float32),
                'prompt_box': np.clip(box + noise, 0, s - 1).astype(np.float32), 'prompt_mask': mask}
