"""Darknet-tiny / 19 / 53 on the MI355X HIP kernels -- drop-in for reference SimpleAICV/classification/backbones/darknet.py
(factories :437-452, ConvBnActBlock :36-68, Darknet19Block :71-113, Darknet53Block :116-146, the three networks :149-434).

Same class names, constructor arguments, parameter / buffer names and registration order (`conv1.layer.0.weight`,
`layer3.Darknet19Block.1.layer.1.running_var`, `block3.7.conv.0.layer.0.weight`, `fc.bias` ...) and the same initialisation
draw order (kaiming-normal fan_out over the convolutions in module order, :251-258), so equal seeds give equal weights and
reference checkpoints load.

Execution: every ConvBnActBlock is the fused conv -> BatchNorm node of the ResNet path (ops.conv_bn_act; ReLU rides in its
BatchNorm-apply kernel); the activations it does not know -- nn.LeakyReLU(0.1) (the default here) and nn.SiLU -- are one more
streaming pass (ops.act, csrc/elemwise.hip).  Max pooling, the global average pool and the classifier are the ResNet kernels;
Darknet-53's residual join is ops.scale_add.  Activations stay NHWC in the compute dtype from the packed input to the pool.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops

__all__ = [
    'darknettiny',
    'darknet19',
    'darknet53',
]

_ACTS = {'silu': lambda: nn.SiLU(inplace=True), 'relu': lambda: nn.ReLU(inplace=True),
         'leakyrelu': lambda: nn.LeakyReLU(0.1, inplace=True)}


class ActivationBlock(nn.Module):
    """Holds the activation's type and slope (no parameters); applied by the enclosing ConvBnActBlock."""

    def __init__(self, act_type='leakyrelu', inplace=True):
        super(ActivationBlock, self).__init__()
        if act_type not in _ACTS:
            raise AssertionError('Unsupport activation function!')
        self.act_type = act_type
        self.act = _ACTS[act_type]()

    def forward(self, x):
        if self.act_type == 'relu':
            return ops.act(x, 'relu')
        return ops.act(x, self.act_type, 0.1 if self.act_type == 'leakyrelu' else 0.)


class ConvBnActBlock(nn.Module):

    def __init__(self, inplanes, planes, kernel_size, stride, padding, groups=1, has_bn=True, has_act=True,
                 act_type='leakyrelu'):
        super(ConvBnActBlock, self).__init__()
        if groups != 1:
            raise NotImplementedError('grouped convolution is not used by the Darknet family')
        self.layer = nn.Sequential(
            nn.Conv2d(inplanes, planes, kernel_size, stride=stride, padding=padding, groups=groups, bias=not has_bn),
            nn.BatchNorm2d(planes) if has_bn else nn.Sequential(),
            ActivationBlock(act_type=act_type, inplace=True) if has_act else nn.Sequential(),
        )
        self.geometry = (stride, padding)
        self.has_bn, self.has_act = has_bn, has_act
        self.act_type = act_type

    def forward(self, x):
        conv, bn, activation = self.layer[0], self.layer[1], self.layer[2]
        stride, padding = self.geometry
        # ReLU after a BatchNorm rides in the BatchNorm-apply kernel; every other activation is its own streaming pass
        fused = self.has_bn and self.has_act and self.act_type == 'relu'
        if self.has_bn:
            x = ops.conv_bn_act(x, conv.weight, bn, stride, padding, fused)
        else:
            x = ops.conv2d(x, conv.weight, conv.bias, stride, padding)
        return activation(x) if (self.has_act and not fused) else x


class Darknet19Block(nn.Module):
    """layer_num alternating 3x3 (inplanes -> planes) / 1x1 (planes -> inplanes) blocks, optional 2x2 max pool"""

    def __init__(self, inplanes, planes, layer_num, use_maxpool=False, act_type='leakyrelu'):
        super(Darknet19Block, self).__init__()
        self.use_maxpool = use_maxpool
        self.Darknet19Block = nn.Sequential(*[
            ConvBnActBlock(inplanes, planes, kernel_size=3, stride=1, padding=1, act_type=act_type) if i % 2 == 0 else
            ConvBnActBlock(planes, inplanes, kernel_size=1, stride=1, padding=0, act_type=act_type) for i in range(layer_num)])
        if use_maxpool:
            self.MaxPool = nn.MaxPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        x = self.Darknet19Block(x)
        return ops.max_pool2d(x, 2, 2, 0) if self.use_maxpool else x


class Darknet53Block(nn.Module):
    """1x1 squeeze to half the channels, 3x3 back, residual join"""

    def __init__(self, inplanes, act_type='leakyrelu'):
        super(Darknet53Block, self).__init__()
        half = int(inplanes // 2)
        self.conv = nn.Sequential(ConvBnActBlock(inplanes, half, kernel_size=1, stride=1, padding=0, act_type=act_type),
                                  ConvBnActBlock(half, inplanes, kernel_size=3, stride=1, padding=1, act_type=act_type))

    def forward(self, x):
        return ops.scale_add(x, self.conv(x))


def _init_like_reference(model):
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)


def _pooled_logits(x, fc):
    x = ops.global_avg_pool(x)
    return ops.linear(x, fc.weight, fc.bias, out_f32=True)


class DarknetTiny(nn.Module):

    def __init__(self, act_type='leakyrelu', num_classes=1000):
        super(DarknetTiny, self).__init__()
        self.num_classes = num_classes
        widths = [3, 16, 32, 64, 128, 256, 512]
        for i in range(1, 7):
            setattr(self, f'conv{i}', ConvBnActBlock(widths[i - 1], widths[i], kernel_size=3, stride=1, padding=1, act_type=act_type))
            if i < 6:
                setattr(self, f'maxpool{i}', nn.MaxPool2d(kernel_size=2, stride=2))
        self.zeropad = nn.ZeroPad2d((0, 1, 0, 1))
        self.maxpool6 = nn.MaxPool2d(kernel_size=2, stride=1)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, self.num_classes)
        _init_like_reference(self)

    def forward(self, x):
        x = ops.pack_input(x)
        for i in range(1, 6):
            x = ops.max_pool2d(getattr(self, f'conv{i}')(x), 2, 2, 0)
        x = self.conv6(x)
        # zeros on the right / bottom edge take part in the stride-1 2x2 maximum (reference :237-238): a real padded tensor
        x = ops.max_pool2d(F.pad(x, (0, 1, 0, 1)), 2, 1, 0)
        return _pooled_logits(x, self.fc)


class Darknet19(nn.Module):

    def __init__(self, act_type='leakyrelu', num_classes=1000):
        super(Darknet19, self).__init__()
        self.num_classes = num_classes
        self.layer1 = ConvBnActBlock(3, 32, kernel_size=3, stride=1, padding=1, act_type=act_type)
        self.maxpool1 = nn.MaxPool2d(kernel_size=2, stride=2)
        for idx, (cin, cout, depth, pool) in enumerate([(32, 64, 1, True), (64, 128, 3, True), (128, 256, 3, True),
                                                        (256, 512, 5, True), (512, 1024, 5, False)]):
            setattr(self, f'layer{idx + 2}', Darknet19Block(cin, cout, layer_num=depth, use_maxpool=pool, act_type=act_type))
        self.layer7 = ConvBnActBlock(1024, self.num_classes, kernel_size=1, stride=1, padding=0, has_bn=False, has_act=False,
                                     act_type=act_type)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        _init_like_reference(self)

    def forward(self, x):
        x = ops.max_pool2d(self.layer1(ops.pack_input(x)), 2, 2, 0)
        for idx in range(2, 8):
            x = getattr(self, f'layer{idx}')(x)
        return ops.global_avg_pool(x).float()


class Darknet53(nn.Module):

    def __init__(self, act_type='leakyrelu', num_classes=1000):
        super(Darknet53, self).__init__()
        self.num_classes = num_classes
        self.conv1 = ConvBnActBlock(3, 32, kernel_size=3, stride=1, padding=1, act_type=act_type)
        widths, repeats = [32, 64, 128, 256, 512, 1024], [1, 2, 8, 8, 4]
        for i, n in enumerate(repeats):
            setattr(self, f'conv{i + 2}', ConvBnActBlock(widths[i], widths[i + 1], kernel_size=3, stride=2, padding=1, act_type=act_type))
            setattr(self, f'block{i + 1}', self.make_layer(inplanes=widths[i + 1], num_blocks=n, act_type=act_type))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(1024, self.num_classes)
        _init_like_reference(self)

    def forward(self, x):
        x = self.conv1(ops.pack_input(x))
        for i in range(1, 6):
            x = getattr(self, f'block{i}')(getattr(self, f'conv{i + 1}')(x))
        return _pooled_logits(x, self.fc)

    def make_layer(self, inplanes, num_blocks, act_type):
        return nn.Sequential(*[Darknet53Block(inplanes, act_type=act_type) for _ in range(num_blocks)])


def darknettiny(**kwargs):
    return DarknetTiny(**kwargs)


def darknet19(**kwargs):
    return Darknet19(**kwargs)


def darknet53(**kwargs):
    return Darknet53(**kwargs)
