"""DINOv3 ConvNeXt (tiny / small / base / large) as a multi-scale detection backbone on the MI355X HIP kernels -- drop-in for reference
SimpleAICV/detection/models/backbones/dinov3convnext.py (LayerNorm2d :27, DropPathBlock :44, Block :80, Dinov3ConvNeXtBackbone :120,
factories :218-252).

Same class / attribute names (`downsample_layers.0.0.weight`, `downsample_layers.2.0.bias`, `stages.2.5.dwconv.weight`,
`stages.0.1.pwconv1.weight`, `stages.3.0.gamma` ...), registration order and initialisation (trunc-normal std 0.02 over convolutions
and linears in module order, :178-182): equal seeds give equal weights, reference checkpoints load.

Execution on one NHWC activation in the compute dtype -- the reference's NCHW <-> NHWC permutes around the token half of every block
disappear:
  4x4 stride-4 stem, 2x2 stride-2 downsampling   -> implicit-GEMM convolution, bias in the epilogue (ops.conv2d)
  LayerNorm2d (over channels, per pixel)          -> the row LayerNorm kernels on the [N*H*W, C] rows (csrc/tfm.hip)
  7x7 depthwise convolution                       -> csrc/dwconv.hip
  LayerNorm -> pwconv1 + GELU -> pwconv2          -> ONE autograd node on the rows (ops_tfm.norm_mlp_branch: GELU and gelu' in the fc1
                                                     GEMM epilogue, gelu' applied in the fc2 data-gradient epilogue)
  input + drop_path(gamma * branch)               -> csrc/elemwise.hip scale_add (gamma's gradient included), per-sample drop factor"""
import numpy as np
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ..... import ops, ops_tfm
from ....classification.backbones.vit import DropPathBlock as _DropPathScale
from ....classification.common import load_state_dict

__all__ = [
    'dinov3convnexttinybackbone',
    'dinov3convnextsmallbackbone',
    'dinov3convnextbasebackbone',
    'dinov3convnextlargebackbone',
]


class LayerNorm2d(nn.Module):
    """LayerNorm over the channel axis of an NCHW-shaped tensor (biased variance, eps inside the sqrt)."""

    def __init__(self, inplanes, eps=1e-6):
        super(LayerNorm2d, self).__init__()
        self.weight = nn.Parameter(torch.ones(inplanes))
        self.bias = nn.Parameter(torch.zeros(inplanes))
        self.eps = eps

    def forward(self, x):
        t = ops._nhwc(x).permute(0, 2, 3, 1)            # NHWC view of a channels-last tensor: no copy
        return ops_tfm.layer_norm(t, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class DropPathBlock(_DropPathScale):

    def forward(self, x):
        w = self.sample_scale(x.shape[0], x.device)
        return x if w is None else ops.sample_scale(x, w)


class _Conv(nn.Conv2d):
    """nn.Conv2d parameter layout, implicit-GEMM execution."""

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0])


class Block(nn.Module):

    def __init__(self, inplanes, drop_path_prob=0.):
        super(Block, self).__init__()
        self.dwconv = nn.Conv2d(inplanes, inplanes, kernel_size=7, padding=3, groups=inplanes)
        self.norm = nn.LayerNorm(inplanes, eps=1e-6)
        self.pwconv1 = nn.Linear(inplanes, 4 * inplanes)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * inplanes, inplanes)
        self.gamma = nn.Parameter(1e-6 * torch.ones((inplanes)), requires_grad=True)
        # if test model,drop_path must set to 0.
        self.drop_path = DropPathBlock(drop_path_prob) if drop_path_prob > 0.0 else nn.Identity()

    def forward(self, x):
        f = ops.depthwise_conv2d(x, self.dwconv.weight, self.dwconv.bias, 1, 3, 1)
        b, c, h, w = f.shape
        rows = ops._nhwc(f).permute(0, 2, 3, 1).reshape(b, h * w, c)
        rows = ops_tfm.norm_mlp_branch(rows, self.norm, self.pwconv1, self.pwconv2)
        f = rows.view(b, h, w, c).permute(0, 3, 1, 2)
        # the reference scales by gamma first and drops after; both are per-element factors, so the order is free
        return ops.scale_add(x, self.drop_path(f), self.gamma)


class Dinov3ConvNeXtBackbone(nn.Module):

    def __init__(self, inplanes=3, embedding_planes=[96, 192, 384, 768], block_nums=[3, 3, 9, 3], drop_path_prob=0.,
                 use_gradient_checkpoint=False):
        super(Dinov3ConvNeXtBackbone, self).__init__()
        assert len(embedding_planes) == len(block_nums)
        self.block_nums = block_nums
        self.use_gradient_checkpoint = use_gradient_checkpoint
        widths = list(embedding_planes)
        layers = [nn.Sequential(_Conv(inplanes, widths[0], kernel_size=4, stride=4), LayerNorm2d(widths[0], eps=1e-6))]
        layers += [nn.Sequential(LayerNorm2d(widths[i], eps=1e-6), _Conv(widths[i], widths[i + 1], kernel_size=2, stride=2))
                   for i in range(len(block_nums) - 1)]
        self.downsample_layers = nn.ModuleList(layers)
        rates = list(np.linspace(0, drop_path_prob, sum(block_nums)))
        first = np.cumsum([0] + list(block_nums))
        self.stages = nn.ModuleList([nn.Sequential(*[Block(inplanes=widths[i], drop_path_prob=rates[first[i] + j])
                                                     for j in range(block_nums[i])]) for i in range(len(block_nums))])
        self.out_channels = widths[:4]
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m.groups == 1:
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x):
        outs = []
        x = ops.pack_input(x)
        for down, stage in zip(self.downsample_layers, self.stages):
            if self.use_gradient_checkpoint:
                x = checkpoint(stage, checkpoint(down, x, use_reentrant=False), use_reentrant=False)
            else:
                x = stage(down(x))
            outs.append(x)
        return outs


def _dinov3convnextbackbone(block_nums, embedding_planes, pretrained_path='', **kwargs):
    model = Dinov3ConvNeXtBackbone(block_nums=block_nums, embedding_planes=embedding_planes, **kwargs)
    if pretrained_path:
        load_state_dict(pretrained_path, model)
    else:
        print('no backbone pretrained model!')
    return model


def dinov3convnexttinybackbone(pretrained_path='', **kwargs):
    return _dinov3convnextbackbone([3, 3, 9, 3], [96, 192, 384, 768], pretrained_path=pretrained_path, **kwargs)


def dinov3convnextsmallbackbone(pretrained_path='', **kwargs):
    return _dinov3convnextbackbone([3, 3, 27, 3], [96, 192, 384, 768], pretrained_path=pretrained_path, **kwargs)


def dinov3convnextbasebackbone(pretrained_path='', **kwargs):
    return _dinov3convnextbackbone([3, 3, 27, 3], [128, 256, 512, 1024], pretrained_path=pretrained_path, **kwargs)


def dinov3convnextlargebackbone(pretrained_path='', **kwargs):
    return _dinov3convnextbackbone([3, 3, 27, 3], [192, 384, 768, 1536], pretrained_path=pretrained_path, **kwargs)
