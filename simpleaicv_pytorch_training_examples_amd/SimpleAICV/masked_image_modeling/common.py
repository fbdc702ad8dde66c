"""Collater of the MAE pre-training loop -- reference SimpleAICV/masked_image_modeling/common.py:16-56
(MAESelfSupervisedPretrainCollater): a batch of HWC float images -> {'image': [B, 3, H, W] float32 (the NHWC-strided
view the reference's permute leaves), 'label': [B, L, p*p*3] float32} where label row (h, w) holds the pixels of patch
(h, w) in (p, q, c) order and, with norm_label, is standardised per patch by its own mean and UNBIASED variance
(`(x - mean) / (var + 1e-4) ** 0.5`, common.py:47-51).  Host code, as in the reference (it runs in the loader's worker
processes); the regression target never touches the model, so nothing of it needs a kernel."""
import numpy as np
import torch

from ..classification.common import load_state_dict  # noqa: F401  (re-exported as the reference module does, common.py:13)


class MAESelfSupervisedPretrainCollater:

    def __init__(self, image_size=224, patch_size=16, norm_label=True):
        assert image_size % patch_size == 0
        self.patch_size = patch_size
        self.patch_nums = image_size // patch_size
        self.norm_label = norm_label

    def __call__(self, data):
        images = torch.from_numpy(np.array([s['image'] for s in data], dtype=np.float32))
        images = images.permute(0, 3, 1, 2).float()                      # B H W 3 -> B 3 H W (a view, like the reference's)
        n, p = self.patch_nums, self.patch_size
        labels = images.reshape(images.shape[0], 3, n, p, n, p)          # (copies: the view is not contiguous in this order)
        labels = torch.einsum('nchpwq->nhwpqc', labels).reshape(images.shape[0], n * n, p * p * 3).float()
        if self.norm_label:
            mean = labels.mean(dim=-1, keepdim=True)
            var = labels.var(dim=-1, keepdim=True)
            labels = ((labels - mean) / (var + 1e-4) ** 0.5).float()
        return {'image': images, 'label': labels}
