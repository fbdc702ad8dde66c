"""Plain (multi-scale) ResNet detection backbone on the MI355X HIP kernels -- the backbone of the reference's RetinaNet /
FCOS families ("other backbones that reuse the same blocks").

Interface contract (reference SimpleAICV/detection/models/backbones/resnet.py:27-282): ResNetBackbone(block, layer_nums,
inplanes, use_gradient_checkpoint) returning [C2, C3, C4, C5], `out_channels`, factories resnet{18,34,50,101,152}backbone
(pretrained_path, **kwargs); it imports ConvBnActBlock / BasicBlock / Bottleneck from the classification backbone (:15),
so parameter / buffer names (`conv1.layer.*`, `layerN.M.convK.layer.*`, `downsample_conv.layer.*`) and the kaiming
fan_out initialisation draw order (:76-83) are those of the classification ResNet.  Execution is the DETR backbone's
(fused implicit-GEMM conv + BN statistics epilogue, NHWC, compute dtype) without the position embedding."""
from .detr_resnet import BasicBlock, Bottleneck, DetrResNetBackbone
from ....classification.common import load_state_dict

__all__ = [
    'resnet18backbone',
    'resnet34backbone',
    'resnet50backbone',
    'resnet101backbone',
    'resnet152backbone',
]


class ResNetBackbone(DetrResNetBackbone):
    """Same stem / stages / outputs as the reference class of this name; the module tree is built by the shared base."""


def _resnetbackbone(block, layers, inplanes, pretrained_path='', **kwargs):
    model = ResNetBackbone(block, layers, inplanes, **kwargs)
    if pretrained_path:
        load_state_dict(pretrained_path, model)
    else:
        print('no backbone pretrained model!')
    return model


def resnet18backbone(pretrained_path='', **kwargs):
    return _resnetbackbone(BasicBlock, [2, 2, 2, 2], 64, pretrained_path=pretrained_path, **kwargs)


def resnet34backbone(pretrained_path='', **kwargs):
    return _resnetbackbone(BasicBlock, [3, 4, 6, 3], 64, pretrained_path=pretrained_path, **kwargs)


def resnet50backbone(pretrained_path='', **kwargs):
    return _resnetbackbone(Bottleneck, [3, 4, 6, 3], 64, pretrained_path=pretrained_path, **kwargs)


def resnet101backbone(pretrained_path='', **kwargs):
    return _resnetbackbone(Bottleneck, [3, 4, 23, 3], 64, pretrained_path=pretrained_path, **kwargs)


def resnet152backbone(pretrained_path='', **kwargs):
    return _resnetbackbone(Bottleneck, [3, 8, 36, 3], 64, pretrained_path=pretrained_path, **kwargs)
