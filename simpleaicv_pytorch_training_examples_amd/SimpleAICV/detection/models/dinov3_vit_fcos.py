"""FCOS on a DINOv3 ViT trunk (reference SimpleAICV/detection/models/dinov3_vit_fcos.py:28-101, factories :109-144): ViT feature map
-> VitPyramidNeck -> levels 1..3 -> RetinaFPN with P5 (P3..P7) -> the shared class / box / centre-ness tower (GroupNorm), box
distances scaled by exp(scales[level]).  Same constructor, module tree (`backbone`, `neck`, `fpn`, `clsregcnt_head`, `scales`),
construction order and output contract as the reference class."""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import backbones
from .backbones.dinov3vit import VitPyramidNeck
from .fpn import RetinaFPN
from .head import FCOSClsRegCntHead

_TRUNKS = ('small', 'small_plus', 'base', 'large', 'large_plus', 'huge_plus')
__all__ = [f'dinov3_vit_{t}_patch16_fcos' for t in _TRUNKS]


class FCOS(nn.Module):

    def __init__(self, backbone_type, backbone_pretrained_path='', planes=256, num_classes=80, use_gradient_checkpoint=False):
        super(FCOS, self).__init__()
        self.planes, self.num_classes = planes, num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.backbone = backbones.__dict__[backbone_type](**{'pretrained_path': backbone_pretrained_path,
                                                             'use_gradient_checkpoint': use_gradient_checkpoint})
        self.neck = VitPyramidNeck(inplanes=self.backbone.out_channels, planes=planes)
        self.fpn = RetinaFPN([planes, planes, planes], planes, use_p5=True)
        self.clsregcnt_head = FCOSClsRegCntHead(planes, num_classes, num_layers=4, use_gn=True, cnt_on_reg=True)
        self.scales = nn.Parameter(torch.tensor([1., 1., 1., 1., 1.], dtype=torch.float32))

    def forward(self, inputs):
        features = self.neck(self.backbone(inputs))[1:4]
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


def _factory(trunk):
    def build(backbone_pretrained_path='', **kwargs):
        return _fcos(f'dinov3_vit_{trunk}_patch16_backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)

    build.__name__ = f'dinov3_vit_{trunk}_patch16_fcos'
    return build


for _t in _TRUNKS:
    globals()[f'dinov3_vit_{_t}_patch16_fcos'] = _factory(_t)
