"""RetinaNet on the MI355X kernels (reference SimpleAICV/detection/models/retinanet.py:27-139):
multi-scale ResNet backbone (C3..C5) -> RetinaFPN (P3..P7) -> class / box towers shared by the five levels.  Same
constructor, module tree, construction order (identical initial weights under the same seed) and output contract:
[cls_heads, reg_heads], per level [B, H, W, anchors, classes] probabilities (fp32) and [B, H, W, anchors, 4] offsets.
Activations are NHWC in memory, so the reference's `permute(0, 2, 3, 1).contiguous()` is a view here."""
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import backbones
from .fpn import RetinaFPN
from .head import RetinaClsHead, RetinaRegHead

__all__ = ['resnet18_retinanet', 'resnet34_retinanet', 'resnet50_retinanet', 'resnet101_retinanet', 'resnet152_retinanet']


class RetinaNet(nn.Module):

    def __init__(self, backbone_type, backbone_pretrained_path='', planes=256, num_anchors=9, num_classes=80,
                 use_gradient_checkpoint=False):
        super(RetinaNet, self).__init__()
        self.planes, self.num_anchors, self.num_classes = planes, num_anchors, num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.backbone = backbones.__dict__[backbone_type](**{'pretrained_path': backbone_pretrained_path,
                                                             'use_gradient_checkpoint': use_gradient_checkpoint})
        self.fpn = RetinaFPN(self.backbone.out_channels[1:4], planes, use_p5=False)
        self.cls_head = RetinaClsHead(planes, num_anchors, num_classes, num_layers=4)
        self.reg_head = RetinaRegHead(planes, num_anchors, num_layers=4)

    @staticmethod
    def _per_anchor(x, last):
        x = x.permute(0, 2, 3, 1).contiguous()             # NHWC memory: no copy
        return x.view(x.shape[0], x.shape[1], x.shape[2], -1, last)

    def forward(self, inputs):
        features = self.backbone(inputs)[1:4]
        features = checkpoint(self.fpn, features, use_reentrant=False) if self.use_gradient_checkpoint else self.fpn(features)
        cls_heads = [self._per_anchor(self.cls_head(f), self.num_classes) for f in features]
        reg_heads = [self._per_anchor(self.reg_head(f), 4) for f in features]
        return [cls_heads, reg_heads]


def _retinanet(backbone_type, backbone_pretrained_path, **kwargs):
    return RetinaNet(backbone_type, backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet18_retinanet(backbone_pretrained_path='', **kwargs):
    return _retinanet('resnet18backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet34_retinanet(backbone_pretrained_path='', **kwargs):
    return _retinanet('resnet34backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet50_retinanet(backbone_pretrained_path='', **kwargs):
    return _retinanet('resnet50backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet101_retinanet(backbone_pretrained_path='', **kwargs):
    return _retinanet('resnet101backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet152_retinanet(backbone_pretrained_path='', **kwargs):
    return _retinanet('resnet152backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)
