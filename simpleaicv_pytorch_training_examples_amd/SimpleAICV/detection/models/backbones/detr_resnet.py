"""DETR ResNet backbone + sine position embedding on the MI355X HIP kernels -- drop-in for the reference.

Interface contract (reference SimpleAICV/detection/models/backbones/detr_resnet.py): PositionEmbeddingBlock
(:28), ConvBnActBlock (:117), BasicBlock (:146), Bottleneck (:195), DetrResNetBackbone (:256) returning
[C2, C3, C4, C5], factories detr_resnet{18,34,50,101,152}backbone (:355-392); identical parameter / buffer
names (`conv1.layer.*`, `layerN.M.convK.layer.*`, `downsample_conv.layer.*`), init draw order and
`out_channels`.  The blocks are the classification ones (the reference duplicates them verbatim): fused
implicit-GEMM conv + BatchNorm statistics epilogue + BN / ReLU / residual kernels, NHWC, compute dtype.
"""
import math

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ..... import ops
from ....._lib import check, lib, ptr, require_gpu, stream
from ....classification.backbones.resnet import BasicBlock, Bottleneck, ConvBnActBlock, _init_like_reference
from ....classification.common import load_state_dict

__all__ = [
    'detr_resnet18backbone',
    'detr_resnet34backbone',
    'detr_resnet50backbone',
    'detr_resnet101backbone',
    'detr_resnet152backbone',
]


class PositionEmbeddingBlock(nn.Module):
    """Sine position embedding of the unpadded region; [B, H, W] bool mask (True = padding) -> [B, 2*inplanes, H, W] fp32.
    One kernel (`saicv_detr_sine_pe`, csrc/input.hip): per sample the running counts of un-padded pixels per column / row are
    normalised to [0, 2 pi] and expanded to sin / cos features (reference detr_resnet.py:28-64)."""

    def __init__(self, inplanes=128, temperature=10000, eps=1e-6):
        super(PositionEmbeddingBlock, self).__init__()
        self.inplanes = inplanes
        self.temperature = temperature
        self.eps = eps
        self.scale = 2 * math.pi

    def forward(self, masks):
        assert masks is not None
        require_gpu(masks)
        m = masks.contiguous()
        m = m if m.dtype in (torch.bool, torch.uint8) else (m != 0)
        b, h, w = m.shape
        out = torch.empty((b, 2 * self.inplanes, h, w), dtype=torch.float32, device=m.device)
        check(lib().saicv_detr_sine_pe(ptr(m), ptr(out), b, h, w, self.inplanes, float(self.temperature), float(self.eps), stream()),
              'detr_sine_pe')
        return out


class DetrResNetBackbone(nn.Module):

    def __init__(self, block, layer_nums, inplanes=64, use_gradient_checkpoint=False):
        super(DetrResNetBackbone, self).__init__()
        self.block = block
        self.layer_nums = layer_nums
        self.inplanes = inplanes
        self.planes = [inplanes, inplanes * 2, inplanes * 4, inplanes * 8]
        self.expansion = 1 if block is BasicBlock else 4
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.conv1 = ConvBnActBlock(3, self.inplanes, kernel_size=7, stride=2, padding=3, groups=1, has_bn=True,
                                    has_act=True)
        self.maxpool1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self.make_layer(self.block, self.planes[0], self.layer_nums[0], stride=1)
        self.layer2 = self.make_layer(self.block, self.planes[1], self.layer_nums[1], stride=2)
        self.layer3 = self.make_layer(self.block, self.planes[2], self.layer_nums[2], stride=2)
        self.layer4 = self.make_layer(self.block, self.planes[3], self.layer_nums[3], stride=2)
        self.out_channels = [self.planes[0] * self.expansion, self.planes[1] * self.expansion,
                             self.planes[2] * self.expansion, self.planes[3] * self.expansion]
        _init_like_reference(self)

    def make_layer(self, block, planes, layer_nums, stride):
        layers = []
        for i in range(0, layer_nums):
            layers.append(block(self.inplanes, planes, stride if i == 0 else 1))
            self.inplanes = planes * self.expansion
        return nn.Sequential(*layers)

    def forward(self, x):
        x = ops.pack_stem_input(x, self.conv1.layer[0])  # compute dtype; the 7x7 stride-2 stem takes a space-to-depth image
        mp = self.maxpool1
        if ops.STEM_POOL_FUSE and isinstance(mp.kernel_size, int) and mp.kernel_size <= 2 * mp.stride + 1 and mp.dilation == 1 and not mp.ceil_mode:
            x = self.conv1(x, pool=(mp.kernel_size, mp.stride, mp.padding))       # conv -> [BN + ReLU + MaxPool as one pass]
        else:
            x = self.conv1(x)
            x = ops.max_pool2d(x, mp.kernel_size, mp.stride, mp.padding)
        outs = []
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = checkpoint(stage, x, use_reentrant=False) if self.use_gradient_checkpoint else stage(x)
            outs.append(x)
        return outs


def _detrresnetbackbone(block, layers, inplanes, pretrained_path='', **kwargs):
    model = DetrResNetBackbone(block, layers, inplanes, **kwargs)
    if pretrained_path:
        load_state_dict(pretrained_path, model)
    else:
        print('no backbone pretrained model!')
    return model


def detr_resnet18backbone(pretrained_path='', **kwargs):
    return _detrresnetbackbone(BasicBlock, [2, 2, 2, 2], 64, pretrained_path=pretrained_path, **kwargs)


def detr_resnet34backbone(pretrained_path='', **kwargs):
    return _detrresnetbackbone(BasicBlock, [3, 4, 6, 3], 64, pretrained_path=pretrained_path, **kwargs)


def detr_resnet50backbone(pretrained_path='', **kwargs):
    return _detrresnetbackbone(Bottleneck, [3, 4, 6, 3], 64, pretrained_path=pretrained_path, **kwargs)


def detr_resnet101backbone(pretrained_path='', **kwargs):
    return _detrresnetbackbone(Bottleneck, [3, 4, 23, 3], 64, pretrained_path=pretrained_path, **kwargs)


def detr_resnet152backbone(pretrained_path='', **kwargs):
    return _detrresnetbackbone(Bottleneck, [3, 8, 36, 3], 64, pretrained_path=pretrained_path, **kwargs)
