"""ConvFormer s18 / s36 / m36 / b36 as a multi-scale detection backbone on the MI355X HIP kernels -- drop-in for reference
SimpleAICV/detection/models/backbones/convformer.py (MetaFormerBackbone :29-117, factories :135-167).

The reference class imports Downsampling / MetaFormerBlock from the classification ConvFormer (:17) and builds the same
`downsample_layers` / `stages` lists without the pooling head: keys, initialisation draws (trunc-normal std 0.02 over convolutions
and linears in module order, :91-97) and kernels are those of classification/backbones/convformer.py; forward returns the four stage
outputs (strides 4 / 8 / 16 / 32) as NHWC activations in the compute dtype."""
import torch.nn as nn

from ....classification.backbones import convformer as _cf
from ....classification.common import load_state_dict

__all__ = [
    'convformers18backbone',
    'convformers36backbone',
    'convformerm36backbone',
    'convformerb36backbone',
]


class MetaFormerBackbone(nn.Module):

    def __init__(self, inplanes=3, embedding_planes=[64, 128, 320, 512], block_nums=[2, 2, 6, 2], dropout_prob=0., drop_path_prob=0.,
                 use_gradient_checkpoint=False):
        super(MetaFormerBackbone, self).__init__()
        assert len(embedding_planes) == len(block_nums)
        self.block_nums = block_nums
        self.use_gradient_checkpoint = use_gradient_checkpoint
        _cf._register_stages(self, inplanes, embedding_planes, block_nums, dropout_prob, drop_path_prob)
        self.out_channels = list(embedding_planes[:4])
        _cf._init_like_reference(self)

    def forward(self, x):
        return _cf._stage_outputs(self, x)


def _metaformerbackbone(block_nums, embedding_planes, pretrained_path='', **kwargs):
    model = MetaFormerBackbone(block_nums=block_nums, embedding_planes=embedding_planes, **kwargs)
    if pretrained_path:
        load_state_dict(pretrained_path, model)
    else:
        print('no backbone pretrained model!')
    return model


def convformers18backbone(pretrained_path='', **kwargs):
    return _metaformerbackbone([3, 3, 9, 3], [64, 128, 320, 512], pretrained_path=pretrained_path, **kwargs)


def convformers36backbone(pretrained_path='', **kwargs):
    return _metaformerbackbone([3, 12, 18, 3], [64, 128, 320, 512], pretrained_path=pretrained_path, **kwargs)


def convformerm36backbone(pretrained_path='', **kwargs):
    return _metaformerbackbone([3, 12, 18, 3], [96, 192, 384, 576], pretrained_path=pretrained_path, **kwargs)


def convformerb36backbone(pretrained_path='', **kwargs):
    return _metaformerbackbone([3, 12, 18, 3], [128, 256, 512, 768], pretrained_path=pretrained_path, **kwargs)
