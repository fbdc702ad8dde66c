"""VAN b0..b6 as a multi-scale detection backbone on the MI355X HIP kernels -- drop-in for reference
SimpleAICV/detection/models/backbones/van.py (VANBackbone :32-130, factories :148-199).

The reference class imports OverlapPatchEmbed / Block from the classification VAN (:18) and registers patch_embed{i} / block{i} /
norm{i} exactly as the classifier does, minus the pooling head -- so the tree, the state_dict keys, the initialisation draws
(:92-105: convolutions in module order) and the kernels are those of classification/backbones/van.py; forward returns the four
normalised stage outputs [C2, C3, C4, C5] (strides 4 / 8 / 16 / 32) as NHWC activations in the compute dtype, which is what the
FPNs consume."""
import torch.nn as nn

from ....classification.backbones import van as _van
from ....classification.common import load_state_dict

__all__ = [
    'vanb0backbone',
    'vanb1backbone',
    'vanb2backbone',
    'vanb3backbone',
    'vanb4backbone',
    'vanb5backbone',
    'vanb6backbone',
]


class VANBackbone(nn.Module):

    def __init__(self, inplanes=3, embedding_planes=[64, 128, 256, 512], mlp_ratios=[4, 4, 4, 4], block_nums=[3, 4, 6, 3],
                 dropout_prob=0., drop_path_prob=0., use_gradient_checkpoint=False):
        super(VANBackbone, self).__init__()
        assert len(embedding_planes) == len(mlp_ratios) == len(block_nums)
        self.block_nums = block_nums
        self.use_gradient_checkpoint = use_gradient_checkpoint
        _van._register_stages(self, inplanes, embedding_planes, mlp_ratios, block_nums, dropout_prob, drop_path_prob)
        self.out_channels = list(embedding_planes[:4])
        _van._init_like_reference(self)

    def forward(self, x):
        return _van._stage_outputs(self, x)


def _vanbackbone(embedding_planes, mlp_ratios, block_nums, pretrained_path='', **kwargs):
    model = VANBackbone(embedding_planes=embedding_planes, mlp_ratios=mlp_ratios, block_nums=block_nums, **kwargs)
    if pretrained_path:
        load_state_dict(pretrained_path, model)
    else:
        print('no backbone pretrained model!')
    return model


def _factory(name):
    widths, depths = _van._VARIANTS['van_' + name[3:5]]

    def build(pretrained_path='', **kwargs):
        return _vanbackbone(widths, [8, 8, 4, 4], depths, pretrained_path=pretrained_path, **kwargs)

    build.__name__ = name
    return build


for _name in __all__:
    globals()[_name] = _factory(_name)
