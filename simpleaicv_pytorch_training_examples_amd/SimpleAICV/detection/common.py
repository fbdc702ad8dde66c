"""Batch collaters of the detection pipelines -- drop-ins for the reference DetectionCollater (SimpleAICV/detection/common.py:
243-288; RetinaNet / FCOS: image canvas + padded annotations) and DETRDetectionCollater (:291-363).

Device-side input contract it defines: image [B, 3, S, S] fp32 as a `.permute(0, 3, 1, 2)` VIEW of an NHWC
batch (channels-last memory, NCHW shape -- exactly what the conv kernels stream), mask [B, S, S] bool with
True on padding, annots [B, max_annots_num, 5] xyxy + class padded with -1, scaled_annots the same boxes as
cxcywh normalised by each image's own (scaled) width / height, plus the host-side scale / size arrays.
S = resize ('yolo_style') or round(resize * 1333 / 800) ('retina_style'); images sit at the top-left.
"""
import numpy as np
import torch

from ..classification.common import load_state_dict  # noqa: F401  (re-exported, as in the reference)


class DETRDetectionCollater:

    def __init__(self, resize=800, resize_type='yolo_style', max_annots_num=100):
        assert resize_type in ['retina_style', 'yolo_style']
        self.resize = resize
        if resize_type == 'retina_style':
            self.resize = int(round(self.resize * 1333. / 800))
        self.max_annots_num = max_annots_num

    def __call__(self, data):
        n, s, m = len(data), self.resize, self.max_annots_num
        canvas = np.zeros((n, s, s, 3), dtype=np.float32)
        masks = torch.ones((n, s, s), dtype=torch.bool)
        annots = np.full((n, m, 5), -1, dtype=np.float32)
        scaled = np.full((n, m, 5), -1, dtype=np.float32)
        scaled_sizes = []
        for i, sample in enumerate(data):
            image, a = sample['image'], sample['annots']
            h, w = image.shape[0], image.shape[1]
            scaled_sizes.append([h, w])
            canvas[i, 0:h, 0:w, :] = image
            masks[i, 0:h, 0:w] = False
            if a.shape[0] > 0:
                annots[i, :a.shape[0], :] = a
                cxcywh = np.concatenate([(a[:, 0:2] + a[:, 2:4]) / 2, a[:, 2:4] - a[:, 0:2]], axis=1)
                cxcywh = cxcywh / np.array([w, h, w, h], dtype=np.float32)
                scaled[i, :a.shape[0], :] = np.concatenate([cxcywh, a[:, 4:5]], axis=1)
        return {
            'image': torch.from_numpy(canvas).permute(0, 3, 1, 2).float(),       # B H W 3 -> B 3 H W (view)
            'annots': torch.from_numpy(annots).float(),
            'mask': masks,
            'scale': np.array([x['scale'] for x in data], dtype=np.float32),
            'size': np.array([x['size'] for x in data], dtype=np.float32),
            'scaled_annots': torch.from_numpy(scaled).float(),
            'scaled_size': np.array(scaled_sizes, dtype=np.float32),
        }


def pad_mask_on_device(scaled_size, resize, device):
    """The DETR padding mask built where it is used: [B, S, S] bool, True outside each image's (h, w) top-left rectangle -- what
    DETRDetectionCollater fills on the host (reference detection/common.py:315-322) and the loop would otherwise copy to the device
    (S * S bytes per image, 8 MiB per batch of eight 1024-pixel canvases) -- from the [B, 2] `scaled_size` array the collater
    returns anyway.  Two broadcast compares; `config.device_pad_mask = True` makes train_detection use it."""
    hw = torch.as_tensor(scaled_size, dtype=torch.float32).to(device, non_blocking=True)
    idx = torch.arange(resize, device=device, dtype=torch.float32)
    return (idx.view(1, -1, 1) >= hw[:, 0].view(-1, 1, 1)) | (idx.view(1, 1, -1) >= hw[:, 1].view(-1, 1, 1))


class DetectionCollater:
    """images at the top-left of a zero [B, S, S, 3] canvas handed over as its NCHW view, annotations [B, max_annots_num, 5]
    (xyxy + class) padded with -1 rows, per-image scale / size arrays (reference :243-288)."""

    def __init__(self, resize=800, resize_type='retina_style', max_annots_num=100):
        assert resize_type in ['retina_style', 'yolo_style']
        self.resize = resize
        if resize_type == 'retina_style':
            self.resize = int(round(self.resize * 1333. / 800))
        self.max_annots_num = max_annots_num

    def __call__(self, data):
        n, s, m = len(data), self.resize, self.max_annots_num
        canvas = np.zeros((n, s, s, 3), dtype=np.float32)
        annots = np.full((n, m, 5), -1, dtype=np.float32)
        for i, sample in enumerate(data):
            image, a = sample['image'], sample['annots']
            canvas[i, 0:image.shape[0], 0:image.shape[1], :] = image
            if a.shape[0] > 0:
                annots[i, :a.shape[0], :] = a
        return {
            'image': torch.from_numpy(canvas).permute(0, 3, 1, 2).float(),
            'annots': torch.from_numpy(annots).float(),
            'scale': np.array([x['scale'] for x in data], dtype=np.float32),
            'size': np.array([x['size'] for x in data], dtype=np.float32),
        }
