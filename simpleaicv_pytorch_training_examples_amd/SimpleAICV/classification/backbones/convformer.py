"""ConvFormer (MetaFormer with separable-convolution token mixers) s18 / s36 / m36 / b36 on the MI355X HIP kernels -- drop-in
for reference SimpleAICV/classification/backbones/convformer.py (Downsampling :16, SepConv :47, Mlp :80, DropPathBlock :103,
MetaFormerBlock :137, MetaFormer :165, factories :264-294).

Same class / attribute names (`downsample_layers.1.pre_norm.weight`, `stages.2.4.token_mixer.pwconv1.weight`,
`stages.0.1.mlp.fc2.weight`, `head.bias` ...), registration order and initialisation (trunc-normal std 0.02 over convolutions and
linears in module order, :229-235): equal seeds give equal weights, reference checkpoints load.

Execution: the reference hops between NCHW (BatchNorm, depthwise convolution) and NHWC (the bias-free nn.Linear layers) with
permutes; here one NHWC activation in the compute dtype serves both -- the linears ARE pointwise GEMMs over its rows
(ops_tfm.linear_nd on the [N, H, W, C] view), the 7x7 depthwise convolution is csrc/dwconv.hip, BatchNorm on block inputs is
ops.batch_norm2d, ReLU / residual joins are csrc/elemwise.hip.  No layout change between the packed input and the pool.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .... import ops, ops_tfm
from .vit import DropPathBlock as _DropPathScale

__all__ = [
    'convformer_s18',
    'convformer_s36',
    'convformer_m36',
    'convformer_b36',
]


def _rows_linear(fc, x):
    """nn.Linear over the channels of an NCHW-shaped, NHWC-strided activation"""
    y = ops_tfm.linear_nd(ops._nhwc(x).permute(0, 2, 3, 1), fc.weight, fc.bias)
    return y.permute(0, 3, 1, 2)


def _norm(module, x):
    return ops.batch_norm2d(x, module) if isinstance(module, nn.BatchNorm2d) else x


class Downsampling(nn.Module):

    def __init__(self, inplanes, planes, kernel_size, stride=1, padding=0, pre_norm=False, post_norm=False):
        super(Downsampling, self).__init__()
        self.conv = nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=stride, padding=padding, bias=True)
        self.pre_norm = nn.BatchNorm2d(inplanes) if pre_norm else nn.Identity()
        self.post_norm = nn.BatchNorm2d(planes) if post_norm else nn.Identity()

    def forward(self, x):
        x = _norm(self.pre_norm, x)
        x = ops.conv2d(x, self.conv.weight, self.conv.bias, self.conv.stride[0], self.conv.padding[0])
        return _norm(self.post_norm, x)


class SepConv(nn.Module):
    """pointwise expand -> ReLU -> k x k depthwise -> pointwise project"""

    def __init__(self, inplanes, kernel_size=7, padding=3, expand_ratio=2):
        super(SepConv, self).__init__()
        middle_planes = int(expand_ratio * inplanes)
        self.pwconv1 = nn.Linear(inplanes, middle_planes, bias=False)
        self.act1 = nn.ReLU(inplace=True)
        self.dwconv = nn.Conv2d(middle_planes, middle_planes, kernel_size=kernel_size, padding=padding, groups=middle_planes, bias=False)
        self.act2 = nn.Identity()
        self.pwconv2 = nn.Linear(middle_planes, inplanes, bias=False)

    def forward(self, x):
        x = ops.act(_rows_linear(self.pwconv1, x), 'relu')
        x = ops.depthwise_conv2d(x, self.dwconv.weight, None, 1, self.dwconv.padding[0], 1)
        return _rows_linear(self.pwconv2, x)


class Mlp(nn.Module):

    def __init__(self, inplanes, mlp_ratio=4, dropout_prob=0.):
        super(Mlp, self).__init__()
        hidden_planes = int(mlp_ratio * inplanes)
        self.fc1 = nn.Linear(inplanes, hidden_planes, bias=False)
        self.act = nn.ReLU(inplace=True)
        self.drop1 = nn.Dropout(dropout_prob)
        self.fc2 = nn.Linear(hidden_planes, inplanes, bias=False)
        self.drop2 = nn.Dropout(dropout_prob)

    def forward(self, x):
        x = ops.act(_rows_linear(self.fc1, x), 'relu')
        if self.drop1.p > 0.:
            x = self.drop1(x)
        x = _rows_linear(self.fc2, x)
        return self.drop2(x) if self.drop2.p > 0. else x


class DropPathBlock(_DropPathScale):

    def forward(self, x):
        w = self.sample_scale(x.shape[0], x.device)
        return x if w is None else ops.sample_scale(x, w)


class MetaFormerBlock(nn.Module):

    def __init__(self, inplanes, dropout_prob=0., drop_path_prob=0.):
        super(MetaFormerBlock, self).__init__()
        self.norm1 = nn.BatchNorm2d(inplanes)
        self.token_mixer = SepConv(inplanes=inplanes, kernel_size=7, padding=3, expand_ratio=2)
        self.norm2 = nn.BatchNorm2d(inplanes)
        self.mlp = Mlp(inplanes=inplanes, mlp_ratio=4, dropout_prob=dropout_prob)
        # if test model,drop_path must set to 0.
        self.drop_path = DropPathBlock(drop_path_prob) if drop_path_prob > 0. else nn.Identity()

    def forward(self, x):
        for norm, fn in ((self.norm1, self.token_mixer), (self.norm2, self.mlp)):
            x = ops.scale_add(x, self.drop_path(fn(ops.batch_norm2d(x, norm))))
        return x


def _register_stages(net, inplanes, embedding_planes, block_nums, dropout_prob, drop_path_prob):
    """downsample_layers / stages in the reference's registration order (convformer.py:185-222; the detection backbone builds the
    same tree, detection/models/backbones/convformer.py:43-84)."""
    widths = [inplanes] + list(embedding_planes)
    # stem: 7x7 stride 4 (padding 2) with a BatchNorm after it; later stages: BatchNorm, then 3x3 stride 2
    net.downsample_layers = nn.ModuleList([
        Downsampling(widths[i], widths[i + 1], kernel_size=7 if i == 0 else 3, stride=4 if i == 0 else 2, padding=2 if i == 0 else 1,
                     pre_norm=i > 0, post_norm=i == 0) for i in range(len(block_nums))])
    rates = list(np.linspace(0, drop_path_prob, sum(block_nums)))
    offsets = np.cumsum([0] + list(block_nums))
    net.stages = nn.ModuleList([
        nn.Sequential(*[MetaFormerBlock(inplanes=embedding_planes[i], dropout_prob=dropout_prob, drop_path_prob=rates[offsets[i] + j])
                        for j in range(block_nums[i])]) for i in range(len(block_nums))])


def _init_like_reference(net):
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    for m in net.modules():
        if isinstance(m, nn.Conv2d) and m.groups == 1:
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)


def _stage_outputs(net, x):
    """Every stage's output (NHWC, compute dtype): the classifier pools the last, the detection backbone returns all four."""
    outs = []
    x = ops.pack_input(x)
    for down, stage in zip(net.downsample_layers, net.stages):
        if net.use_gradient_checkpoint:
            x = checkpoint(stage, checkpoint(down, x, use_reentrant=False), use_reentrant=False)
        else:
            x = stage(down(x))
        outs.append(x)
    return outs


class MetaFormer(nn.Module):

    def __init__(self, inplanes=3, embedding_planes=[64, 128, 320, 512], block_nums=[2, 2, 6, 2], dropout_prob=0., drop_path_prob=0.,
                 num_classes=1000, use_gradient_checkpoint=False):
        super(MetaFormer, self).__init__()
        assert len(embedding_planes) == len(block_nums)
        self.block_nums = block_nums
        self.num_classes = num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        _register_stages(self, inplanes, embedding_planes, block_nums, dropout_prob, drop_path_prob)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.head = nn.Linear(embedding_planes[3], num_classes)
        _init_like_reference(self)

    def forward(self, x):
        x = ops.global_avg_pool(_stage_outputs(self, x)[-1])
        return ops.linear(x, self.head.weight, self.head.bias, out_f32=True)


def _metaformer(block_nums, embedding_planes, **kwargs):
    return MetaFormer(block_nums=block_nums, embedding_planes=embedding_planes, **kwargs)


def convformer_s18(**kwargs):
    return _metaformer([3, 3, 9, 3], [64, 128, 320, 512], **kwargs)


def convformer_s36(**kwargs):
    return _metaformer([3, 12, 18, 3], [64, 128, 320, 512], **kwargs)


def convformer_m36(**kwargs):
    return _metaformer([3, 12, 18, 3], [96, 192, 384, 576], **kwargs)


def convformer_b36(**kwargs):
    return _metaformer([3, 12, 18, 3], [128, 256, 512, 768], **kwargs)
