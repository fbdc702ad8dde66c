"""Host-side sample transforms and the batch collater of the SAM pipeline -- drop-in for the reference.

Interface contract (reference SimpleAICV/interactive_segmentation/common.py): SamResize (:18),
SamRandomHorizontalFlip (:55), SamNormalize (:99), SAMBatchCollater (:129-232), load_state_dict.
The collater defines the device-side input contract: image [B, 3, S, S] fp32 in TRUE NCHW (per-sample
permute, then stack), mask [B, 1, S, S], prompt_point [B, P, 3], prompt_box [B, 4], prompt_mask
[B, 1, S/4, S/4]; zero padding to the square canvas at the top-left.
OpenCV is optional here: nearest-neighbour resizing (all the collater needs) is done in numpy with
OpenCV's index rule floor(dst * src / dst_size); SamResize's image interpolation needs cv2.
"""
import numpy as np
import torch

from ..classification.common import load_state_dict  # noqa: F401  (re-exported, as in the reference)


def resize_nearest(arr, width, height):
    """cv2.resize(arr, (width, height), interpolation=cv2.INTER_NEAREST) for 2-d / 3-d arrays."""
    h, w = arr.shape[:2]
    ys = np.minimum((np.arange(height) * (h / height)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(width) * (w / width)).astype(np.int64), w - 1)
    return arr[ys][:, xs]


class SamResize:

    def __init__(self, resize=1024):
        self.resize = resize

    def __call__(self, sample):
        import cv2   # bilinear image resampling; not needed by the synthetic generators
        image = sample['image']
        h, w, _ = image.shape
        factor = self.resize / max(h, w)
        resize_h, resize_w = int(round(h * factor)), int(round(w * factor))
        sample['image'] = cv2.resize(image, (resize_w, resize_h))
        sample['box'][0:4] *= factor
        sample['mask'] = resize_nearest(sample['mask'], resize_w, resize_h)
        sample['size'] = np.array([resize_h, resize_w]).astype(np.float32)
        sample['prompt_point'][:, 0:2] *= factor
        sample['prompt_box'][0:4] *= factor
        sample['prompt_mask'] = resize_nearest(sample['prompt_mask'], resize_w, resize_h)
        return sample


class SamRandomHorizontalFlip:

    def __init__(self, prob=0.5):
        self.prob = prob

    def __call__(self, sample):
        if np.random.uniform(0, 1) < self.prob:
            image, box, prompt_box, prompt_point = sample['image'], sample['box'], sample['prompt_box'], sample['prompt_point']
            sample['image'] = image[:, ::-1, :]
            sample['mask'] = sample['mask'][:, ::-1]
            sample['prompt_mask'] = sample['prompt_mask'][:, ::-1]
            _, w, _ = image.shape
            x1, x2 = box[0].copy(), box[2].copy()
            box[0], box[2] = w - x2, w - x1
            x1, x2 = prompt_box[0].copy(), prompt_box[2].copy()
            prompt_box[0], prompt_box[2] = w - x2, w - x1
            for i in range(len(prompt_point)):
                prompt_point[i][0] = w - prompt_point[i][0]
        return sample


class SamNormalize:

    def __init__(self, mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]):
        self.mean = np.expand_dims(np.expand_dims(np.array(mean), axis=0), axis=0)
        self.std = np.expand_dims(np.expand_dims(np.array(std), axis=0), axis=0)

    def __call__(self, sample):
        sample['image'] = (sample['image'] - self.mean) / self.std
        return sample


class SAMBatchCollater:

    def __init__(self, resize):
        self.resize = resize
        assert resize % 64 == 0
        self.prompt_mask_size = resize // 4

    def __call__(self, data):
        s = self.resize
        input_images, input_boxes, input_masks, input_prompt_masks = [], [], [], []
        for x in data:
            image = x['image']
            canvas = np.zeros((s, s, 3), dtype=np.float32)
            canvas[0:image.shape[0], 0:image.shape[1], :] = image
            input_images.append(torch.from_numpy(canvas).permute(2, 0, 1))          # [3, H, W] view, stacked below
            box = np.zeros((4), dtype=np.float32)
            box[0:x['box'].shape[0]] = x['box'][0:4]
            input_boxes.append(torch.from_numpy(box))
            mask = np.zeros((s, s), dtype=np.float32)
            mask[0:x['mask'].shape[0], 0:x['mask'].shape[1]] = x['mask']
            input_masks.append(torch.from_numpy(mask))
            pm = x['prompt_mask']
            h, w = pm.shape
            factor = self.prompt_mask_size / max(h, w)
            resize_h, resize_w = int(round(h * factor)), int(round(w * factor))
            pm = resize_nearest(pm, resize_w, resize_h)
            pcanvas = np.zeros((self.prompt_mask_size, self.prompt_mask_size), dtype=np.float32)
            pcanvas[0:pm.shape[0], 0:pm.shape[1]] = pm
            input_prompt_masks.append(torch.from_numpy(pcanvas))
        return {
            'image': torch.stack(input_images, dim=0).float(),
            'box': torch.stack(input_boxes, dim=0).float(),
            'mask': torch.stack(input_masks, dim=0).unsqueeze(1).float(),
            'size': np.array([x['size'] for x in data], dtype=np.float32),
            'prompt_point': torch.stack([torch.from_numpy(x['prompt_point']) for x in data], dim=0).float(),
            'prompt_box': torch.stack([torch.from_numpy(x['prompt_box']) for x in data], dim=0).float(),
            'prompt_mask': torch.stack(input_prompt_masks, dim=0).unsqueeze(1).float(),
        }
