"""CIFAR ResNets (3x3 stride-1 stem, no max-pool) on the MI355X HIP kernels.

Mirrors reference SimpleAICV/classification/backbones/resnetforcifar.py: class ResNetCifar
(:27-99) and factories resnet18cifar..resnet152cifar (:108-125); blocks and state_dict keys are
shared with resnet.py exactly as the reference shares them (:16).
"""
from .... import ops
from .resnet import BasicBlock, Bottleneck, ConvBnActBlock, _ResNetBase, _init_like_reference

__all__ = [
    'resnet18cifar',
    'resnet34cifar',
    'resnet50cifar',
    'resnet101cifar',
    'resnet152cifar',
]


class ResNetCifar(_ResNetBase):

    def __init__(self, block, layer_nums, inplanes=64, num_classes=100):
        super(ResNetCifar, self).__init__()
        self._build_trunk(block, layer_nums, inplanes, num_classes)
        self.conv1 = ConvBnActBlock(3, self.inplanes, kernel_size=3, stride=1, padding=1, groups=1, has_bn=True,
                                    has_act=True)
        self._build_stages_and_head()
        _init_like_reference(self)

    def forward(self, x):
        x = ops.pack_input(x)
        x = self.conv1(x)
        x = self._stages(x)
        return self._head(x)


def _resnetcifar(block, layers, inplanes, **kwargs):
    return ResNetCifar(block, layers, inplanes, **kwargs)


def resnet18cifar(**kwargs):
    return _resnetcifar(BasicBlock, [2, 2, 2, 2], 64, **kwargs)


def resnet34cifar(**kwargs):
    return _resnetcifar(BasicBlock, [3, 4, 6, 3], 64, **kwargs)


def resnet50cifar(**kwargs):
    return _resnetcifar(Bottleneck, [3, 4, 6, 3], 64, **kwargs)


def resnet101cifar(**kwargs):
    return _resnetcifar(Bottleneck, [3, 4, 23, 3], 64, **kwargs)


def resnet152cifar(**kwargs):
    return _resnetcifar(Bottleneck, [3, 8, 36, 3], 64, **kwargs)
