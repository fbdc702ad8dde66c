"""Per-step host helpers of the classification path (reference SimpleAICV/classification/
common.py): ClassificationCollater (:645-665), AverageMeter (:668-684), AccMeter (:687-706),
load_state_dict (:758-840), get_amp_type (:843-881) and the Mixup / CutMix collater the ViT configs use (:19).

Only what sits on the training-step path is mirrored; the cv2 / PIL / torchvision transforms
of the reference (dataset side, CPU worker processes) are out of scope.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .mixupcutmixclassificationcollator import MixupCutmixClassificationCollater

__all__ = ['ClassificationCollater', 'MixupCutmixClassificationCollater', 'AverageMeter', 'AccMeter', 'load_state_dict',
           'get_amp_type']


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
