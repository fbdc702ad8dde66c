"""ViT detection backbone on the hot-path blocks -- drop-in for the reference factories
(SimpleAICV/detection/models/backbones/vit.py: VitPyramidNeck :27, PatchEmbeddingBlock :83, ViTBackbone :118,
vit_base_patch16_backbone / vit_large_patch16_backbone / vit_huge_patch14_backbone :217-226).

Same constructor arguments, parameter names / shapes / registration order (`patch_embed.proj.*`, `pos_embed`,
`blocks.N.*`) and initialisation draw order (:168-174), so seeds and checkpoints carry over.  The trunk is the
classification ViT without class token and final norm: patch embedding (implicit-GEMM, NHWC output = token layout),
position embedding, the fused pre-LN TransformerEncoderLayer nodes, then the [B, N, C] tokens handed back as a
[B, C, h, w] map.  VitPyramidNeck's stride-2 2x2 transposed convolutions are GEMMs here: kernel == stride, so
every input pixel owns a disjoint 2x2 output block -- out[b, 2h+i, 2w+j, o] = sum_c x[b, h, w, c] W[c, o, i, j] + bias[o]
is one linear layer with (i, j, o) output columns followed by a depth-to-space regrouping."""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ..... import ops, ops_tfm
from ....classification.backbones.vit import PatchEmbeddingBlock as _PatchEmbed, TransformerEncoderLayer
from ...common import load_state_dict

__all__ = [
    'vit_base_patch16_backbone',
    'vit_large_patch16_backbone',
    'vit_huge_patch14_backbone',
]


def _gelu_nhwc(x):
    """GELU of a [B, C, h, w] map held in NHWC memory, without a layout copy (the kernel is element-wise over memory)."""
    return ops_tfm.gelu(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


def _transposed_conv2x2(x, conv):
    """nn.ConvTranspose2d(kernel 2, stride 2, padding 0) + GELU on a [B, C, h, w] map (NHWC memory) -> [B, O, 2h, 2w]."""
    b, c, h, w = x.shape
    o = conv.out_channels
    tokens = x.permute(0, 2, 3, 1).reshape(b, h * w, c)                       # view for NHWC memory
    wmat = conv.weight.permute(2, 3, 1, 0).reshape(4 * o, c)                  # rows (i, j, o), columns c
    bias = conv.bias.repeat(4) if conv.bias is not None else None
    y = ops_tfm.linear_nd(tokens, wmat, bias)                                 # [B, h*w, (i, j, o)]
    y = ops_tfm.gelu(y)                                                       # element-wise: before or after the regrouping
    y = y.view(b, h, w, 2, 2, o).permute(0, 1, 3, 2, 4, 5).reshape(b, 2 * h, 2 * w, o)
    return y.permute(0, 3, 1, 2)                                              # NCHW-shaped over NHWC memory


class VitPyramidNeck(nn.Module):

    def __init__(self, inplanes, planes):
        super(VitPyramidNeck, self).__init__()
        self.P2 = nn.Sequential(
            nn.ConvTranspose2d(inplanes, planes, kernel_size=2, stride=2, padding=0, output_padding=0, bias=True),
            nn.GELU(),
            nn.ConvTranspose2d(planes, planes, kernel_size=2, stride=2, padding=0, output_padding=0, bias=True),
            nn.GELU(),
        )
        self.P3 = nn.Sequential(
            nn.ConvTranspose2d(inplanes, planes, kernel_size=2, stride=2, padding=0, output_padding=0, bias=True),
            nn.GELU(),
        )
        self.P4 = nn.Sequential(
            nn.Conv2d(inplanes, planes, kernel_size=1, stride=1, padding=0, bias=True),
            nn.GELU(),
        )
        self.P5 = nn.Sequential(
            nn.MaxPool2d(kernel_size=2, stride=2),
            nn.GELU(),
        )

    def forward(self, x):
        P2 = _transposed_conv2x2(_transposed_conv2x2(x, self.P2[0]), self.P2[2])
        P3 = _transposed_conv2x2(x, self.P3[0])
        P4 = _gelu_nhwc(ops.conv2d(x, self.P4[0].weight, self.P4[0].bias, 1, 0))
        P5 = _gelu_nhwc(ops.max_pool2d(P4, 2, 2, 0))
        return [P2, P3, P4, P5]


class PatchEmbeddingBlock(_PatchEmbed):
    """Reference :83-115 returns the tokens together with the [b, c, h, w] shape of the convolution output."""

    def forward(self, x):
        p = self.stride
        tokens = super(PatchEmbeddingBlock, self).forward(x)
        return tokens, [x.shape[0], tokens.shape[-1], x.shape[2] // p, x.shape[3] // p]


class ViTBackbone(nn.Module):

    def __init__(self, patch_size, embedding_planes, block_nums, head_nums, feedforward_ratio, image_size=224,
                 dropout_prob=0., drop_path_prob=0., use_gradient_checkpoint=False):
        super(ViTBackbone, self).__init__()
        self.image_size = image_size
        self.patch_size = patch_size
        self.embedding_planes = embedding_planes
        self.block_nums = block_nums
        self.head_nums = head_nums
        self.feedforward_ratio = feedforward_ratio
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.patch_embed = PatchEmbeddingBlock(3, self.embedding_planes, kernel_size=self.patch_size,
                                               stride=self.patch_size, padding=0, groups=1, has_norm=False)
        self.pos_embed = nn.Parameter(torch.ones(1, (self.image_size // self.patch_size) ** 2, self.embedding_planes))
        self.embedding_dropout = nn.Dropout(dropout_prob)
        rates = [0. if drop_path_prob == 0. else drop_path_prob * (i / (self.block_nums - 1)) for i in range(self.block_nums)]
        self.blocks = nn.ModuleList([
            TransformerEncoderLayer(self.embedding_planes, self.head_nums, feedforward_ratio=self.feedforward_ratio,
                                    dropout_prob=dropout_prob, drop_path_prob=rates[i]) for i in range(self.block_nums)])
        self.out_channels = embedding_planes
        for m in self.modules():                       # reference :168-174, same draw order
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.trunc_normal_(self.pos_embed, std=.02)

    def forward(self, x):
        x, [b, c, h, w] = self.patch_embed(x)
        x = x + self.pos_embed.to(x.dtype)
        x = self.embedding_dropout(x)
        for block in self.blocks:
            x = checkpoint(block, x, use_reentrant=False) if self.use_gradient_checkpoint else block(x)
        return x.reshape(b, h, w, c).permute(0, 3, 1, 2)       # [B, C, h, w] over NHWC memory


def _vitbackbone(patch_size, embedding_planes, block_nums, head_nums, feedforward_ratio, pretrained_path='', **kwargs):
    model = ViTBackbone(patch_size, embedding_planes, block_nums, head_nums, feedforward_ratio, **kwargs)
    if pretrained_path:
        load_state_dict(pretrained_path, model)
    else:
        print('no backbone pretrained model!')
    return model


def vit_base_patch16_backbone(**kwargs):
    return _vitbackbone(16, 768, 12, 12, 4, **kwargs)


def vit_large_patch16_backbone(**kwargs):
    return _vitbackbone(16, 512, 24, 16, 4, **kwargs)


def vit_huge_patch14_backbone(**kwargs):
    return _vitbackbone(14, 1280, 32, 16, 4, **kwargs)
