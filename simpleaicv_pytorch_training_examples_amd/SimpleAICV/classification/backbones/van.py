"""Visual Attention Network (VAN b0..b6) on the MI355X HIP kernels -- drop-in for reference
SimpleAICV/classification/backbones/van.py (DWConv :21, Mlp :39, LKA :60, Attention :96, DropPathBlock :118, Block :153,
OverlapPatchEmbed :190, VAN :212, factories :317-370).

Same class / attribute names (`block2.1.attn.spatial_gating_unit.conv_spatial.weight`, `block1.0.layer_scale_1`,
`patch_embed3.norm.running_mean`, `head.weight` ...), registration order and initialisation draws (:274-286: trunc-normal
linears, normal(0, sqrt(2 / fan_out)) convolutions in module order), so equal seeds give equal weights and reference
checkpoints load.

Execution on NHWC activations in the compute dtype:
  1x1 / patch-embedding convolutions  -> implicit-GEMM kernels (ops.conv2d, bias in the epilogue)
  5x5, 7x7 dilation 3, 3x3 depthwise  -> csrc/dwconv.hip (ops.depthwise_conv2d)
  BatchNorm2d on block inputs         -> statistics pass + the fused blocks' finalize / apply / backward kernels (ops.batch_norm2d)
  ReLU, u * attn, x + layer_scale * f -> csrc/elemwise.hip (ops.act / ops.mul / ops.scale_add, the scale's gradient included)
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .... import ops
from .vit import DropPathBlock as _DropPathScale

__all__ = [
    'van_b0',
    'van_b1',
    'van_b2',
    'van_b3',
    'van_b4',
    'van_b5',
    'van_b6',
]


def _pointwise(conv, x):
    return ops.conv2d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0])


def _depthwise(conv, x):
    return ops.depthwise_conv2d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], conv.dilation[0])


class DWConv(nn.Module):

    def __init__(self, inplanes=768):
        super(DWConv, self).__init__()
        self.dwconv = nn.Conv2d(inplanes, inplanes, kernel_size=3, stride=1, padding=1, bias=True, groups=inplanes)

    def forward(self, x):
        return _depthwise(self.dwconv, x)


class Mlp(nn.Module):
    """1x1 expand -> 3x3 depthwise -> ReLU -> 1x1 project (dropout modules kept for the attribute names)"""

    def __init__(self, inplanes, hidden_planes, planes, dropout_prob=0.):
        super(Mlp, self).__init__()
        self.fc1 = nn.Conv2d(inplanes, hidden_planes, 1)
        self.dwconv = DWConv(hidden_planes)
        self.act = nn.ReLU(inplace=True)
        self.fc2 = nn.Conv2d(hidden_planes, planes, 1)
        self.drop = nn.Dropout(dropout_prob)

    def forward(self, x):
        x = ops.act(self.dwconv(_pointwise(self.fc1, x)), 'relu')
        x = _pointwise(self.fc2, self.drop(x) if self.drop.p > 0. else x)
        return self.drop(x) if self.drop.p > 0. else x


class LKA(nn.Module):
    """large-kernel attention: 5x5 depthwise -> 7x7 depthwise at dilation 3 -> 1x1, gating its own input"""

    def __init__(self, inplanes):
        super(LKA, self).__init__()
        self.conv0 = nn.Conv2d(inplanes, inplanes, kernel_size=5, stride=1, padding=2, groups=inplanes, bias=True)
        self.conv_spatial = nn.Conv2d(inplanes, inplanes, kernel_size=7, stride=1, padding=9, groups=inplanes, dilation=3, bias=True)
        self.conv1 = nn.Conv2d(inplanes, inplanes, kernel_size=1, stride=1, padding=0, bias=True)

    def forward(self, x):
        gate = _pointwise(self.conv1, _depthwise(self.conv_spatial, _depthwise(self.conv0, x)))
        return ops.mul(x, gate)


class Attention(nn.Module):

    def __init__(self, inplanes):
        super(Attention, self).__init__()
        self.proj_1 = nn.Conv2d(inplanes, inplanes, 1)
        self.activation = nn.ReLU(inplace=True)
        self.spatial_gating_unit = LKA(inplanes)
        self.proj_2 = nn.Conv2d(inplanes, inplanes, 1)

    def forward(self, x):
        y = self.spatial_gating_unit(ops.act(_pointwise(self.proj_1, x), 'relu'))
        return ops.scale_add(x, _pointwise(self.proj_2, y))


class DropPathBlock(_DropPathScale):
    """stochastic depth per sample; the factor (0 or 1 / keep) multiplies the branch in one streaming pass"""

    def forward(self, x):
        w = self.sample_scale(x.shape[0], x.device)
        return x if w is None else ops.sample_scale(x, w)


class Block(nn.Module):

    def __init__(self, inplanes, mlp_ratio=4., dropout_prob=0., drop_path_prob=0.):
        super(Block, self).__init__()
        self.norm1 = nn.BatchNorm2d(inplanes)
        self.attn = Attention(inplanes)
        self.norm2 = nn.BatchNorm2d(inplanes)
        self.mlp = Mlp(inplanes=inplanes, hidden_planes=int(inplanes * mlp_ratio), planes=inplanes, dropout_prob=dropout_prob)
        self.layer_scale_1 = nn.Parameter(1e-5 * torch.ones((1, inplanes, 1, 1)), requires_grad=True)
        self.layer_scale_2 = nn.Parameter(1e-5 * torch.ones((1, inplanes, 1, 1)), requires_grad=True)
        # if test model,drop_path must set to 0.
        self.drop_path = DropPathBlock(drop_path_prob) if drop_path_prob > 0. else nn.Identity()

    def _branch(self, x, norm, fn, scale):
        f = fn(ops.batch_norm2d(x, norm))
        if isinstance(self.drop_path, DropPathBlock):
            # the reference scales first and drops after; both are per-element factors, so the order is free
            f = self.drop_path(f)
        return ops.scale_add(x, f, scale)

    def forward(self, x):
        x = self._branch(x, self.norm1, self.attn, self.layer_scale_1)
        return self._branch(x, self.norm2, self.mlp, self.layer_scale_2)


class OverlapPatchEmbed(nn.Module):

    def __init__(self, patch_size=7, stride=4, inplanes=3, embedding_planes=768):
        super(OverlapPatchEmbed, self).__init__()
        self.proj = nn.Conv2d(inplanes, embedding_planes, kernel_size=patch_size, stride=stride,
                              padding=(patch_size // 2, patch_size // 2))
        self.norm = nn.BatchNorm2d(embedding_planes)

    def forward(self, x):
        return ops.batch_norm2d(_pointwise(self.proj, x), self.norm)


def _register_stages(net, inplanes, embedding_planes, mlp_ratios, block_nums, dropout_prob, drop_path_prob):
    """patch_embed{i} / block{i} / norm{i} for the four stages, in the reference's registration order (van.py:232-260; the detection
    backbone builds the same tree, detection/models/backbones/van.py:52-80)."""
    rates = list(np.linspace(0, drop_path_prob, sum(block_nums)))
    width_in, first = inplanes, 0
    for i, (width, ratio, depth) in enumerate(zip(embedding_planes, mlp_ratios, block_nums)):
        setattr(net, f'patch_embed{i + 1}', OverlapPatchEmbed(patch_size=7 if i == 0 else 3, stride=4 if i == 0 else 2,
                                                               inplanes=width_in, embedding_planes=width))
        setattr(net, f'block{i + 1}', nn.ModuleList([
            Block(inplanes=width, mlp_ratio=ratio, dropout_prob=dropout_prob, drop_path_prob=rates[first + j]) for j in range(depth)]))
        setattr(net, f'norm{i + 1}', nn.BatchNorm2d(width))
        width_in, first = width, first + depth


def _init_like_reference(net):
    for m in net.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.Conv2d):
            fan_out = (m.kernel_size[0] * m.kernel_size[1] * m.out_channels) // m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()
    for m in net.modules():
        if isinstance(m, nn.Conv2d) and m.groups == 1:
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)


def _stage_outputs(net, x):
    """The normalised output of every stage (NHWC, compute dtype); the classifier pools the last one, the detection backbone
    returns all four."""
    def run(fn, t):
        return checkpoint(fn, t, use_reentrant=False) if net.use_gradient_checkpoint else fn(t)

    outs = []
    x = ops.pack_input(x)
    for i in range(len(net.block_nums)):
        x = run(getattr(net, f'patch_embed{i + 1}'), x)
        for blk in getattr(net, f'block{i + 1}'):
            x = run(blk, x)
        norm = getattr(net, f'norm{i + 1}')
        x = run(lambda t, bn=norm: ops.batch_norm2d(t, bn), x)
        outs.append(x)
    return outs


class VAN(nn.Module):

    def __init__(self, inplanes=3, embedding_planes=[64, 128, 256, 512], mlp_ratios=[4, 4, 4, 4], block_nums=[3, 4, 6, 3],
                 dropout_prob=0., drop_path_prob=0., num_classes=1000, use_gradient_checkpoint=False):
        super(VAN, self).__init__()
        assert len(embedding_planes) == len(mlp_ratios) == len(block_nums)
        self.block_nums = block_nums
        self.num_classes = num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        _register_stages(self, inplanes, embedding_planes, mlp_ratios, block_nums, dropout_prob, drop_path_prob)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.head = nn.Linear(embedding_planes[3], num_classes)
        _init_like_reference(self)

    def forward(self, x):
        x = ops.global_avg_pool(_stage_outputs(self, x)[-1])
        return ops.linear(x, self.head.weight, self.head.bias, out_f32=True)


def _van(embedding_planes, mlp_ratios, block_nums, **kwargs):
    return VAN(embedding_planes=embedding_planes, mlp_ratios=mlp_ratios, block_nums=block_nums, **kwargs)


_VARIANTS = {
    'van_b0': ([32, 64, 160, 256], [3, 3, 5, 2]),
    'van_b1': ([64, 128, 320, 512], [2, 2, 4, 2]),
    'van_b2': ([64, 128, 320, 512], [3, 3, 12, 3]),
    'van_b3': ([64, 128, 320, 512], [3, 5, 27, 3]),
    'van_b4': ([64, 128, 320, 512], [3, 6, 40, 3]),
    'van_b5': ([96, 192, 480, 768], [3, 3, 24, 3]),
    'van_b6': ([96, 192, 384, 768], [6, 6, 90, 6]),
}


def _factory(name):
    widths, depths = _VARIANTS[name]

    def build(**kwargs):
        return _van(embedding_planes=widths, mlp_ratios=[8, 8, 4, 4], block_nums=depths, **kwargs)

    build.__name__ = name
    return build


van_b0, van_b1, van_b2, van_b3 = _factory('van_b0'), _factory('van_b1'), _factory('van_b2'), _factory('van_b3')
van_b4, van_b5, van_b6 = _factory('van_b4'), _factory('van_b5'), _factory('van_b6')
