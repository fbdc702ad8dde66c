"""FCOS on the MI355X kernels (reference SimpleAICV/detection/models/fcos.py:27-127): multi-scale ResNet
backbone (C3..C5) -> RetinaFPN with P6 / P7 grown from P5 -> one class / box / centre-ness head shared by the five levels, a
learnable log-scale per level on the box distances.  Same constructor, module tree, construction order (identical initial weights
under the same seed) and output contract: [cls_heads, reg_heads, center_heads], per level [B, H, W, classes] / [B, H, W, 4] /
[B, H, W, 1].  Activations are NHWC in memory, so the reference's `permute(0, 2, 3, 1).contiguous()` is a view here."""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import backbones
from .fpn import RetinaFPN
from .head import FCOSClsRegCntHead

__all__ = ['resnet18_fcos', 'resnet34_fcos', 'resnet50_fcos', 'resnet101_fcos', 'resnet152_fcos']


class FCOS(nn.Module):

    def __init__(self, backbone_type, backbone_pretrained_path='', planes=256, num_classes=80, use_gradient_checkpoint=False):
        super(FCOS, self).__init__()
        self.planes, self.num_classes = planes, num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.backbone = backbones.__dict__[backbone_type](**{'pretrained_path': backbone_pretrained_path,
                                                             'use_gradient_checkpoint': use_gradient_checkpoint})
        self.fpn = RetinaFPN(self.backbone.out_channels[1:4], planes, use_p5=True)
        self.clsregcnt_head = FCOSClsRegCntHead(planes, num_classes, num_layers=4, use_gn=True, cnt_on_reg=True)
        self.scales = nn.Parameter(torch.tensor([1., 1., 1., 1., 1.], dtype=torch.float32))

    def forward(self, inputs):
        features = self.backbone(inputs)[1:4]
        features = checkpoint(self.fpn, features, use_reentrant=False) if self.use_gradient_checkpoint else self.fpn(features)
        out = ([], [], [])
        for level, feature in enumerate(features):
            cls_out, reg_out, cnt_out = (t.permute(0, 2, 3, 1).contiguous() for t in self.clsregcnt_head(feature))
            out[0].append(cls_out)
            out[1].append(reg_out * torch.exp(self.scales[level]))
            out[2].append(cnt_out)
        return list(out)


def _fcos(backbone_type, backbone_pretrained_path, **kwargs):
    return FCOS(backbone_type, backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet18_fcos(backbone_pretrained_path='', **kwargs):
    return _fcos('resnet18backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet34_fcos(backbone_pretrained_path='', **kwargs):
    return _fcos('resnet34backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet50_fcos(backbone_pretrained_path='', **kwargs):
    return _fcos('resnet50backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet101_fcos(backbone_pretrained_path='', **kwargs):
    return _fcos('resnet101backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet152_fcos(backbone_pretrained_path='', **kwargs):
    return _fcos('resnet152backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)
