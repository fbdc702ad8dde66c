"""Per-step host helpers of the classification path (reference SimpleAICV/classification/
common.py): ClassificationCollater (:645-665), AverageMeter (:668-684), AccMeter (:687-706),
load_state_dict (:758-840), get_amp_type (:843-881) and the Mixup / CutMix collater the ViT configs use (:19).

Of the dataset side (CPU worker processes in the reference) the two steps that can move to the device batch are here
(SURVEY.md section 8f rank 3): the mean / std normalisation of TorchMeanStdNormalize (:228-248) on the uint8 batch
(Uint8ClassificationCollater + normalize_on_device) and RandomErasing (:561-640, host call and plan / erase_on_device).
The PIL policy transforms the fine-tuning configs put in front of them are host code and stay host code: Opencv2PIL (:22-37),
PIL2Opencv (:40-55) and AutoAugment / RandAugment (auto_rand_augment.py, re-exported here as the reference does at :18).
The torchvision-backed Torch* transforms and the cv2 ones stay out of scope (neither library is in the image, so neither their
parameter draws nor their resampling could be pinned).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .auto_rand_augment import AutoAugment, RandAugment
from .mixupcutmixclassificationcollator import MixupCutmixClassificationCollater

__all__ = ['ClassificationCollater', 'MixupCutmixClassificationCollater', 'Uint8ClassificationCollater', 'normalize_on_device',
           'RandomErasing', 'AutoAugment', 'RandAugment', 'Opencv2PIL', 'PIL2Opencv', 'AverageMeter', 'AccMeter', 'load_state_dict',
           'get_amp_type']


class Opencv2PIL:
    """sample['image'] (HWC array, any real dtype in 0..255) -> PIL image, for the PIL policy transforms"""

    def __call__(self, sample):
        from PIL import Image
        sample['image'] = Image.fromarray(np.uint8(sample['image']))
        return sample


class PIL2Opencv:
    """sample['image'] (PIL) -> float32 HWC array, the form the rest of the pipeline and the collaters take"""

    def __call__(self, sample):
        sample['image'] = np.asarray(sample['image']).astype(np.float32)
        return sample


class ClassificationCollater:
    """list of {'image': HWC float array, 'label': int} -> {'image': [B,3,H,W] fp32 that is
    NCHW-shaped but NHWC-strided (the permute is a view), 'label': int64 [B]}.  The HIP stem
    kernel (saicv_pack_input) consumes exactly this layout without a copy."""

    def __init__(self):
        pass

    def __call__(self, data):
        images = torch.from_numpy(np.asarray([s['image'] for s in data], dtype=np.float32))
        labels = torch.from_numpy(np.asarray([s['label'] for s in data], dtype=np.float32)).long()
        return {'image': images.permute(0, 3, 1, 2), 'label': labels}


class Uint8ClassificationCollater:
    """Loader-side half of the f3 path: the worker processes hand over the RAW uint8 HWC images (no
    ToTensor / Normalize on the CPU: reference common.py:228-248 does that per sample) and the batch travels to the device as
    [B, H, W, 3] uint8 -- a quarter of the fp32 batch's pinned-memory and PCIe traffic; `normalize_on_device` then does the
    dataset normalisation on the device.  Labels as ClassificationCollater."""

    def __call__(self, data):
        images = np.stack([np.asarray(s['image'], dtype=np.uint8) for s in data])
        labels = torch.from_numpy(np.array([s['label'] for s in data]).astype(np.float32)).long()
        return {'image': torch.from_numpy(images), 'label': labels}


def normalize_on_device(images_u8, mean, std):
    """uint8 [B, H, W, C] on the device -> float32 [B, C, H, W] as the NHWC-strided view ClassificationCollater hands to the
    loop: ((float)v / 255 - mean[c]) / std[c] with the roundings of torchvision's ToTensor + Normalize, i.e. bit-identical to
    the reference's TorchMeanStdNormalize (common.py:228-248) applied per sample on the host (csrc/input.hip)."""
    from ..._lib import check, lib, ptr, require_gpu, stream
    require_gpu(images_u8)
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.is_contiguous()
    c = images_u8.shape[-1]
    m = torch.as_tensor(mean, dtype=torch.float32).to(images_u8.device)
    s = torch.as_tensor(std, dtype=torch.float32).to(images_u8.device)
    assert m.numel() == c and s.numel() == c
    out = torch.empty(images_u8.shape, dtype=torch.float32, device=images_u8.device)
    check(lib().saicv_u8_normalize(ptr(images_u8), ptr(m), ptr(s), ptr(out), images_u8.numel(), c, stream()), 'u8_normalize')
    return out.permute(0, 3, 1, 2)


class RandomErasing:
    """Random Erasing (reference common.py:561-640; the ViT fine-tuning configs use mode 'pixel', prob 0.25): same constructor,
    same numpy draw ORDER -- `uniform(0, 1) < prob`, the count, then per box up to ten attempts of (area, log-aspect) and, when
    the box fits, (top, left) and the fill -- so a seeded host call erases the same pixels with the same values as the reference
    (pinned by tests/golden/random_erasing.pt, produced by the reference class).

    `plan(h, w, c)` draws one sample's boxes WITHOUT the per-pixel fill of mode 'pixel' (the reference spends h * w * c normal
    draws of the host generator per box there), `erase_on_device(images, plans)` fills them on a [B, H, W, C] fp32 device batch:
    'const' / 'rand' colours come from the plan (bit-identical to the host path), 'pixel' values from a counter-based N(0, 1)
    generator on the device (csrc/input.hip)."""

    def __init__(self, prob=0.25, min_area=0.02, max_area=1 / 3, min_aspect=0.3, max_aspect=None, mode='pixel', min_count=1,
                 max_count=None):
        self.prob, self.min_area, self.max_area = prob, min_area, max_area
        max_aspect = max_aspect if max_aspect else 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))
        assert mode in ['const', 'rand', 'pixel']
        self.mode = mode
        self.min_count = min_count
        self.max_count = max_count if max_count else min_count

    def _boxes(self, image_h, image_w, image_c, fill):
        """the reference's draws for one image -> [(top, left, h, w, fill value or None)]"""
        out = []
        if not np.random.uniform(0, 1) < self.prob:
            return out
        area = image_h * image_w
        count = self.min_count if self.min_count == self.max_count else np.random.randint(self.min_count, self.max_count)
        for _ in range(count):
            for _ in range(10):
                target_area = np.random.uniform(self.min_area, self.max_area) * area / count
                aspect_ratio = math.exp(np.random.uniform(*self.log_aspect_ratio))
                h = int(round(math.sqrt(target_area * aspect_ratio)))
                w = int(round(math.sqrt(target_area / aspect_ratio)))
                if w < image_w and h < image_h:
                    top = np.random.randint(0, image_h - h)
                    left = np.random.randint(0, image_w - w)
                    value = None
                    if self.mode == 'pixel':
                        if fill:
                            value = np.random.normal(loc=0.0, scale=1.0, size=(h, w, image_c))
                    elif self.mode == 'rand':
                        value = np.random.normal(loc=0.0, scale=1.0, size=(1, 1, image_c))
                    else:
                        value = np.zeros((1, 1, image_c), dtype=np.float32)
                    out.append((top, left, h, w, value))
                    break
        return out

    def __call__(self, sample):
        """sample: {'image': [h, w, c] array (modified in place, as the reference does), 'label'}"""
        image = sample['image']
        for top, left, h, w, value in self._boxes(image.shape[0], image.shape[1], image.shape[2], True):
            image[top:top + h, left:left + w, :] = value
        sample['image'] = image
        return sample

    def plan(self, image_h, image_w, image_c=3):
        return self._boxes(image_h, image_w, image_c, False)

    def erase_on_device(self, images, plans, seed=0):
        """images: float32 [B, H, W, C] on the device (contiguous; filled in place); plans: one plan() per sample."""
        from ..._lib import EraseBox, check, lib, ptr, require_gpu, stream
        require_gpu(images)
        assert images.dtype == torch.float32 and images.dim() == 4 and images.is_contiguous() and len(plans) == images.shape[0]
        b, h, w, c = images.shape
        rounds = max((len(p) for p in plans), default=0)
        for r in range(rounds):                               # a sample's boxes may overlap: applied in sequence, as the reference does
            entries = [(i, p[r]) for i, p in enumerate(plans) if len(p) > r]
            arr = (EraseBox * len(entries))()
            for k, (i, (top, left, bh, bw, value)) in enumerate(entries):
                arr[k].b, arr[k].top, arr[k].left, arr[k].h, arr[k].w = i, top, left, bh, bw
                arr[k].mode = 1 if self.mode == 'pixel' else 0
                if value is not None:
                    for ch in range(c):
                        arr[k].color[ch] = float(np.float32(value.reshape(-1)[ch]))
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(images.device, non_blocking=True)
            check(lib().saicv_random_erase(ptr(images), ptr(dev), len(entries), b, h, w, c, (seed * 1000003 + r) & 0xffffffff, stream()),
                  'random_erase')
        return images


class AverageMeter:
    '''Computes and stores the average and current value'''

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class AccMeter:
    '''top-1 / top-5 accuracy from correct counts'''

    def __init__(self):
        self.reset()

    def reset(self):
        self.acc1_correct_num = self.acc5_correct_num = self.sample_num = 0
        self.acc1 = self.acc5 = 0

    def update(self, acc1_correct_num, acc5_correct_num, sample_num):
        self.acc1_correct_num += acc1_correct_num
        self.acc5_correct_num += acc5_correct_num
        self.sample_num += sample_num

    def compute(self):
        self.acc1 = float(self.acc1_correct_num) / self.sample_num if self.sample_num != 0 else 0
        self.acc5 = float(self.acc5_correct_num) / self.sample_num if self.sample_num != 0 else 0


def load_state_dict(saved_model_path, model, excluded_layer_name=(),
                    loading_new_input_size_position_encoding_weight=False):
    '''Name- and shape-filtered non-strict load of a saved model.state_dict(); for ViT a
    position embedding saved at another input size is resized bicubically (reference
    common.py:758-840).  Works on the arena-backed parameters: copies are layout agnostic.'''
    if not saved_model_path:
        print('No pretrained model file!')
        return
    saved = torch.load(saved_model_path, map_location=torch.device('cpu'), weights_only=True)
    own = model.state_dict()
    keep, skipped = {}, []
    for name, weight in saved.items():
        ok = name in own and weight.shape == own[name].shape and not any(e in name for e in excluded_layer_name)
        if ok:
            keep[name] = weight
        else:
            skipped.append(name)
    if (loading_new_input_size_position_encoding_weight and 'pos_embed' not in keep
            and hasattr(model, 'cls_token') and hasattr(model, 'pos_embed')):
        ncls = model.cls_token.shape[1]
        planes = model.pos_embed.shape[2]
        side = int((model.pos_embed.shape[1] - ncls) ** 0.5)
        src_name = next((n for n in saved if 'pos_embed' in n), None)
        if src_name is not None:
            src = saved[src_name]
            src_side = int((src.shape[1] - ncls) ** 0.5)
            grid = src[:, ncls:, :].reshape(-1, src_side, src_side, planes).permute(0, 3, 1, 2)
            grid = F.interpolate(grid, size=(side, side), mode='bicubic').permute(0, 2, 3, 1).flatten(1, 2)
            keep[src_name] = torch.cat((src[:, 0:ncls, :], grid), dim=1)
            if 'pos_embed' in skipped:
                skipped.remove('pos_embed')
    if len(keep) == 0:
        print('No pretrained parameters to load!')
    else:
        print(f'load/model weight nums:{len(keep)}/{len(own)}')
        print(f'not loaded save layer weight:\n{skipped}')
        model.load_state_dict(keep, strict=False)
        from ... import ops
        ops.bump_weights_epoch()


def get_amp_type(model):
    """Autocast dtype for the device the model lives on.  The reference only returns bf16 for a
    hard-coded list of NVIDIA boards (common.py:862-877) and would silently fall back to fp16 on
    an MI355X; here every CDNA3/CDNA4 part (gfx942 / gfx950) -- which have native bf16 MFMA --
    returns bf16, everything else keeps the reference rule."""
    device = next(model.parameters()).device
    if device.type != 'cuda':
        return torch.bfloat16
    props = torch.cuda.get_device_properties(device)
    arch = getattr(props, 'gcnArchName', '') or ''
    if arch.startswith(('gfx95', 'gfx94', 'gfx90a')):
        return torch.bfloat16
    name = torch.cuda.get_device_name(device)
    nvidia_bf16 = ['RTX PRO 6000', 'H20', 'L20', 'L40', '4090', '5090', 'A100', 'A800', 'H100', 'H800']
    if props.major >= 8 and any(n in name for n in nvidia_bf16):
        return torch.bfloat16
    return torch.float16
