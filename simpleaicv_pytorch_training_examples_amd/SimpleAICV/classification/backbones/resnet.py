"""ResNet family on the MI355X HIP kernels -- drop-in for the reference factories.

Interface contract (reference SimpleAICV/classification/backbones/resnet.py):
  * class names ConvBnActBlock (:19), BasicBlock (:51), Bottleneck (:100), ResNet (:158) and
    factories resnet18..resnet152 (:254-271) with the same constructor arguments;
  * identical parameter / buffer names, shapes and registration order, e.g.
    `conv1.layer.0.weight`, `layer1.0.conv1.layer.1.running_mean`, `fc.weight`, so reference
    checkpoints load and `build_optimizer`'s name rules apply unchanged;
  * identical initialisation draw order (kaiming-normal fan_out on conv weights, :206-213),
    so the same seed gives the same initial weights as the reference.

What differs is how forward executes: each ConvBnActBlock is ONE fused autograd node
(implicit-GEMM conv with BN statistics in its epilogue -> BN-apply + residual + ReLU) running
on hand-written gfx950 kernels over NHWC activations; conv weights are kept channels_last
(KRSC in memory) so packed copies and gradients need no transposes.
"""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .... import ops

__all__ = [
    'resnet18',
    'resnet34',
    'resnet50',
    'resnet101',
    'resnet152',
]


class ConvBnActBlock(nn.Module):
    """Conv2d -> BatchNorm2d -> ReLU with the reference's switches (resnet.py:19-48): `has_bn=False` makes the convolution
    biased and drops the normalisation (conv + bias epilogue, then one ReLU pass); `groups == inplanes == planes` is a
    depthwise convolution (csrc/dwconv.hip).  Other group counts are not built."""

    def __init__(self, inplanes, planes, kernel_size, stride, padding, groups=1, has_bn=True, has_act=True):
        super(ConvBnActBlock, self).__init__()
        self.depthwise = groups != 1 and groups == inplanes and groups == planes
        if groups != 1 and not self.depthwise:
            raise NotImplementedError(f'ConvBnActBlock(groups={groups}) with {inplanes} -> {planes} channels: only dense '
                                      '(groups=1) and depthwise (groups == inplanes == planes) convolutions have kernels')
        # nn.Conv2d / nn.BatchNorm2d are used as parameter containers only (state_dict keys
        # layer.0.weight[, layer.0.bias], layer.1.{weight,bias,running_mean,running_var,num_batches_tracked}).
        self.layer = nn.Sequential(
            nn.Conv2d(inplanes, planes, kernel_size, stride=stride, padding=padding, groups=groups, bias=not has_bn),
            nn.BatchNorm2d(planes) if has_bn else nn.Sequential(),
            nn.ReLU(inplace=True) if has_act else nn.Sequential(),
        )
        self.stride = stride
        self.padding = padding
        self.has_bn = has_bn
        self.has_act = has_act

    def forward(self, x, residual=None, act=None, want_skip=False, pool=None, defer=False):
        """`residual` / `act` let the enclosing residual block fuse its add + ReLU in here; `want_skip` also
        returns the input as an alias the block uses for its shortcut, so that the shortcut's gradient is
        added inside this conv's dgrad epilogue (no separate gradient-sum kernel).  `pool` = (kernel, stride, padding) of the
        nn.MaxPool2d that follows the block (the stem): BatchNorm-apply, ReLU and the pooling run as one pass.  `defer`: the block
        is a residual block's shortcut and its output goes ONLY into the `residual` argument of the block's last convolution,
        which then applies this BatchNorm itself (ops.conv_bn_act)."""
        conv = self.layer[0]
        relu = self.has_act if act is None else act
        if self.has_bn and not self.depthwise:
            if defer and not relu and residual is None and not want_skip and pool is None:
                return ops.conv_bn_act(x, conv.weight, self.layer[1], self.stride, self.padding, False, defer=True)
            if pool is not None:
                return ops.conv_bn_act(x, conv.weight, self.layer[1], self.stride, self.padding, relu, pool=pool)
            return ops.conv_bn_act(x, conv.weight, self.layer[1], self.stride, self.padding, relu, residual, want_skip)
        # the unfused forms: [depthwise] convolution (+ bias) -> [BatchNorm] -> [+ residual] -> [ReLU], one kernel each
        if self.depthwise:
            y = ops.depthwise_conv2d(x, conv.weight, conv.bias, self.stride, self.padding)
        else:
            y = ops.conv2d(x, conv.weight, conv.bias, self.stride, self.padding)
        if self.has_bn:
            y = ops.batch_norm2d(y, self.layer[1])
        if residual is not None:
            y = ops.scale_add(residual, y)
        if relu:
            y = ops.act(y, 'relu')
        return (y, x) if want_skip else y


class BasicBlock(nn.Module):

    def __init__(self, inplanes, planes, stride=1):
        super(BasicBlock, self).__init__()
        self.downsample = True if stride != 1 or inplanes != planes * 1 else False
        self.conv1 = ConvBnActBlock(inplanes, planes, kernel_size=3, stride=stride, padding=1, groups=1,
                                    has_bn=True, has_act=True)
        self.conv2 = ConvBnActBlock(planes, planes, kernel_size=3, stride=1, padding=1, groups=1, has_bn=True,
                                    has_act=False)
        self.relu = nn.ReLU(inplace=True)
        if self.downsample:
            self.downsample_conv = ConvBnActBlock(inplanes, planes, kernel_size=1, stride=stride, padding=0,
                                                  groups=1, has_bn=True, has_act=False)

    def forward(self, x):
        out, skip = self.conv1(x, want_skip=True)
        identity = self.downsample_conv(skip, defer=ops.DS_JOIN_FUSE) if self.downsample else skip
        # relu(bn2(conv2(out)) + identity), fused into conv2's BN-apply kernel
        return self.conv2(out, residual=identity, act=True)


class Bottleneck(nn.Module):

    def __init__(self, inplanes, planes, stride=1):
        super(Bottleneck, self).__init__()
        self.downsample = True if stride != 1 or inplanes != planes * 4 else False
        self.conv1 = ConvBnActBlock(inplanes, planes, kernel_size=1, stride=1, padding=0, groups=1, has_bn=True,
                                    has_act=True)
        self.conv2 = ConvBnActBlock(planes, planes, kernel_size=3, stride=stride, padding=1, groups=1,
                                    has_bn=True, has_act=True)
        self.conv3 = ConvBnActBlock(planes, planes * 4, kernel_size=1, stride=1, padding=0, groups=1,
                                    has_bn=True, has_act=False)
        self.relu = nn.ReLU(inplace=True)
        if self.downsample:
            self.downsample_conv = ConvBnActBlock(inplanes, planes * 4, kernel_size=1, stride=stride, padding=0,
                                                  groups=1, has_bn=True, has_act=False)

    def forward(self, x):
        out, skip = self.conv1(x, want_skip=True)
        identity = self.downsample_conv(skip, defer=ops.DS_JOIN_FUSE) if self.downsample else skip
        out = self.conv2(out)
        return self.conv3(out, residual=identity, act=True)


def _init_like_reference(model):
    """Same draw order as reference resnet.py:206-213 so equal seeds give equal weights; then
    conv weights move to channels_last (values unchanged, KRSC in memory)."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)


class _ResNetBase(nn.Module):
    """Shared trunk of ResNet (ImageNet stem) and ResNetCifar (3x3 stem, no max-pool)."""

    def _build_trunk(self, block, layer_nums, inplanes, num_classes):
        self.block = block
        self.layer_nums = layer_nums
        self.num_classes = num_classes
        self.inplanes = inplanes
        self.planes = [inplanes, inplanes * 2, inplanes * 4, inplanes * 8]
        self.expansion = 1 if block is BasicBlock else 4

    def _build_stages_and_head(self):
        self.layer1 = self.make_layer(self.block, self.planes[0], self.layer_nums[0], stride=1)
        self.layer2 = self.make_layer(self.block, self.planes[1], self.layer_nums[1], stride=2)
        self.layer3 = self.make_layer(self.block, self.planes[2], self.layer_nums[2], stride=2)
        self.layer4 = self.make_layer(self.block, self.planes[3], self.layer_nums[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(self.planes[3] * self.expansion, self.num_classes)

    def make_layer(self, block, planes, layer_nums, stride):
        layers = []
        for i in range(0, layer_nums):
            layers.append(block(self.inplanes, planes, stride if i == 0 else 1))
            self.inplanes = planes * self.expansion
        return nn.Sequential(*layers)

    def _stages(self, x, use_checkpoint=False):
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = checkpoint(stage, x, use_reentrant=False) if use_checkpoint else stage(x)
        return x

    def _head(self, x):
        x = ops.global_avg_pool(x)                      # [N, C] in the compute dtype
        return ops.linear(x, self.fc.weight, self.fc.bias, out_f32=True)   # fp32 logits


class ResNet(_ResNetBase):

    def __init__(self, block, layer_nums, inplanes=64, num_classes=1000, use_gradient_checkpoint=False):
        super(ResNet, self).__init__()
        self._build_trunk(block, layer_nums, inplanes, num_classes)
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.conv1 = ConvBnActBlock(3, self.inplanes, kernel_size=7, stride=2, padding=3, groups=1, has_bn=True,
                                    has_act=True)
        self.maxpool1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self._build_stages_and_head()
        _init_like_reference(self)

    def forward(self, x):
        x = ops.pack_stem_input(x, self.conv1.layer[0])  # compute dtype; the 7x7 stride-2 stem takes a space-to-depth image
        mp = self.maxpool1
        if (ops.STEM_POOL_FUSE and self.conv1.has_act and isinstance(mp.kernel_size, int) and isinstance(mp.stride, int)
                and isinstance(mp.padding, int) and mp.kernel_size <= 2 * mp.stride + 1 and mp.dilation == 1 and not mp.ceil_mode):
            x = self.conv1(x, pool=(mp.kernel_size, mp.stride, mp.padding))       # conv -> [BN + ReLU + MaxPool as one pass]
        else:
            x = self.conv1(x)
            x = ops.max_pool2d(x, mp.kernel_size, mp.stride, mp.padding)
        x = self._stages(x, self.use_gradient_checkpoint)
        return self._head(x)


def _resnet(block, layers, inplanes, **kwargs):
    return ResNet(block, layers, inplanes, **kwargs)


def resnet18(**kwargs):
    return _resnet(BasicBlock, [2, 2, 2, 2], 64, **kwargs)


def resnet34(**kwargs):
    return _resnet(BasicBlock, [3, 4, 6, 3], 64, **kwargs)


def resnet50(**kwargs):
    return _resnet(Bottleneck, [3, 4, 6, 3], 64, **kwargs)


def resnet101(**kwargs):
    return _resnet(Bottleneck, [3, 4, 23, 3], 64, **kwargs)


def resnet152(**kwargs):
    return _resnet(Bottleneck, [3, 8, 36, 3], 64, **kwargs)
