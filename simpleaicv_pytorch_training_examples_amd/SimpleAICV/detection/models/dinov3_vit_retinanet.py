"""RetinaNet on a DINOv3 ViT trunk (reference SimpleAICV/detection/models/dinov3_vit_retinanet.py:28-112, factories :120-156): the
single-scale ViT feature map goes through VitPyramidNeck (stride 4 / 8 / 16 / 32 maps of `planes` channels), levels 1..3 feed a
RetinaFPN (P3..P7) and the class / box towers shared by the five levels.  Same constructor, module tree (`backbone`, `neck`, `fpn`,
`cls_head`, `reg_head`), construction order and output contract as the reference class."""
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import backbones
from .backbones.dinov3vit import VitPyramidNeck
from .fpn import RetinaFPN
from .head import RetinaClsHead, RetinaRegHead

_TRUNKS = ('small', 'small_plus', 'base', 'large', 'large_plus', 'huge_plus')
__all__ = [f'dinov3_vit_{t}_patch16_retinanet' for t in _TRUNKS]


class RetinaNet(nn.Module):

    def __init__(self, backbone_type, backbone_pretrained_path='', planes=256, num_anchors=9, num_classes=80,
                 use_gradient_checkpoint=False):
        super(RetinaNet, self).__init__()
        self.planes, self.num_anchors, self.num_classes = planes, num_anchors, num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.backbone = backbones.__dict__[backbone_type](**{'pretrained_path': backbone_pretrained_path,
                                                             'use_gradient_checkpoint': use_gradient_checkpoint})
        self.neck = VitPyramidNeck(inplanes=self.backbone.out_channels, planes=planes)
        self.fpn = RetinaFPN([planes, planes, planes], planes, use_p5=False)
        self.cls_head = RetinaClsHead(planes, num_anchors, num_classes, num_layers=4)
        self.reg_head = RetinaRegHead(planes, num_anchors, num_layers=4)

    @staticmethod
    def _per_anchor(x, last):
        x = x.permute(0, 2, 3, 1).contiguous()             # NHWC memory: no copy
        return x.view(x.shape[0], x.shape[1], x.shape[2], -1, last)

    def forward(self, inputs):
        features = self.neck(self.backbone(inputs))[1:4]
        features = checkpoint(self.fpn, features, use_reentrant=False) if self.use_gradient_checkpoint else self.fpn(features)
        cls_heads = [self._per_anchor(self.cls_head(f), self.num_classes) for f in features]
        reg_heads = [self._per_anchor(self.reg_head(f), 4) for f in features]
        return [cls_heads, reg_heads]


def _retinanet(backbone_type, backbone_pretrained_path, **kwargs):
    return RetinaNet(backbone_type, backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def _factory(trunk):
    def build(backbone_pretrained_path='', **kwargs):
        return _retinanet(f'dinov3_vit_{trunk}_patch16_backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)

    build.__name__ = f'dinov3_vit_{trunk}_patch16_retinanet'
    return build


for _t in _TRUNKS:
    globals()[f'dinov3_vit_{_t}_patch16_retinanet'] = _factory(_t)
