"""DINOv3 ViT detection backbones on the hot-path kernels -- drop-in for the reference factories
(SimpleAICV/detection/models/backbones/dinov3vit.py: PatchEmbed :35, LayerScale :73, Mlp :93, SwiGLUFFN :118,
RopePositionEmbedding :144, LinearKMaskedBias :280, SelfAttention :298, SelfAttentionBlock :356, DinoVisionTransformer :453,
factories :575-721).

Same constructor arguments, parameter / buffer names, shapes and registration order (`patch_embed.proj.*`,
`rope_embed.periods`, `blocks.N.{norm1,attn.qkv(+bias_mask),attn.proj,ls1.gamma,norm2,mlp.{fc1,fc2}|{w1,w2,w3},ls2.gamma}`,
`norm.*`) and the same random-draw order at construction (nn.Conv2d / nn.Linear are used as parameter containers, then the
reference's own re-initialisation), so seeds and DINOv3 checkpoints carry over.  The RoPE coordinate augmentations (shift /
jitter / rescale, training only) draw from the host generator exactly as the reference does.

What runs on the GPU: patch embedding (implicit GEMM, NHWC output = token layout), LayerNorm, the linears (bias epilogue; GELU
fused for the `mlp` form), rotary embedding of q / k in one pass over the packed projection (csrc/elemwise.hip rope_kernel),
streaming attention (head dim 64: every published size except the 7B model, whose head dim is 128 -- refused), the SwiGLU gate
in one pass, LayerScale fused with the residual add (x + gamma * branch)."""
import math
from functools import partial

import numpy as np
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ..... import ops, ops_tfm
from ...common import load_state_dict
from .vit import VitPyramidNeck  # noqa: F401  (re-exported as in the reference, :21: the DINOv3 detectors import it from here)

__all__ = [
    'dinov3_vit_small_patch16_backbone',
    'dinov3_vit_small_plus_patch16_backbone',
    'dinov3_vit_base_patch16_backbone',
    'dinov3_vit_large_patch16_backbone',
    'dinov3_vit_large_plus_patch16_backbone',
    'dinov3_vit_huge_plus_patch16_backbone',
    'dinov3_vit_7b_patch16_backbone',
]


class PatchEmbed(nn.Module):

    def __init__(self, inplanes=3, planes=768, kernel_size=16, stride=16, padding=0, has_norm=False):
        super(PatchEmbed, self).__init__()
        if padding != 0 or kernel_size != stride:
            raise NotImplementedError('patch embedding is a plain stride-p convolution on the hot path')
        self.planes = planes
        self.stride = stride
        self.proj = nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=stride, padding=padding)
        self.norm = nn.LayerNorm(planes, eps=1e-6) if has_norm else nn.Identity()
        self.has_norm = has_norm
        k = 1 / (inplanes * (kernel_size ** 2))
        nn.init.uniform_(self.proj.weight, -math.sqrt(k), math.sqrt(k))
        if self.proj.bias is not None:
            nn.init.uniform_(self.proj.bias, -math.sqrt(k), math.sqrt(k))

    def forward(self, x):
        h, w = x.shape[2] // self.stride, x.shape[3] // self.stride
        x = ops_tfm.patch_embed(x, self.proj.weight, self.proj.bias, self.stride)       # [B, h*w, C]
        if self.has_norm:
            x = ops_tfm.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x.reshape(-1, h, w, self.planes)


class LayerScale(nn.Module):
    """x * gamma (reference :73-90).  Inside a block the scale is fused with the residual add (`_residual`)."""

    def __init__(self, inplanes, init_values=1e-5, inplace=False):
        super(LayerScale, self).__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(torch.empty(inplanes))
        nn.init.constant_(self.gamma, init_values)

    def forward(self, x):
        return _residual(None, x, self.gamma)


def _residual(x, branch, gamma):
    """x + gamma[c] * branch on [B, N, C] tokens (x None: gamma * branch; gamma None: x + branch), one pass"""
    b, n, c = branch.shape
    as4 = lambda t: t.reshape(b, n, 1, c).permute(0, 3, 1, 2)                           # NCHW-shaped view of NHWC memory
    out = ops.scale_add(as4(x) if x is not None else None, as4(branch), gamma)
    return out.permute(0, 2, 3, 1).reshape(b, n, c)


class Mlp(nn.Module):

    def __init__(self, inplanes, hidden_planes, planes, drop_prob=0.0, bias=True):
        super(Mlp, self).__init__()
        if drop_prob != 0.:
            raise NotImplementedError('dropout is 0 in every reference DINOv3 configuration')
        self.fc1 = nn.Linear(inplanes, hidden_planes, bias=bias)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_planes, planes, bias=bias)
        self.drop = nn.Dropout(drop_prob)

    def forward(self, x):
        x = ops_tfm.gelu(ops_tfm.linear_nd(x, self.fc1.weight, self.fc1.bias))
        return ops_tfm.linear_nd(x, self.fc2.weight, self.fc2.bias)


class SwiGLUFFN(nn.Module):

    def __init__(self, inplanes, hidden_planes, planes, drop_prob=0.0, bias=True, align_to=8):
        super(SwiGLUFFN, self).__init__()
        d = int(hidden_planes * 2 / 3)
        swiglu_hidden_planes = d + (-d % align_to)
        self.w1 = nn.Linear(inplanes, swiglu_hidden_planes, bias=bias)
        self.w2 = nn.Linear(inplanes, swiglu_hidden_planes, bias=bias)
        self.w3 = nn.Linear(swiglu_hidden_planes, planes, bias=bias)

    def forward(self, x):
        x1 = ops_tfm.linear_nd(x, self.w1.weight, self.w1.bias)
        x2 = ops_tfm.linear_nd(x, self.w2.weight, self.w2.bias)
        return ops_tfm.linear_nd(ops_tfm.swiglu(x1, x2), self.w3.weight, self.w3.bias)


class RopePositionEmbedding(nn.Module):
    """-> (sin, cos) [H*W, head_planes] fp32 (reference :144-259): axial coordinates in [-1, 1], periods base^(2i / (D/2)) or a
    min / max period range, angles tiled twice; the training-time shift / jitter / rescale draws come from the HOST generator in
    the reference's order.  A few hundred values: built with torch ops on the parameter's device."""

    def __init__(self, embedding_planes, head_nums, base=100.0, min_period=None, max_period=None, normalize_coords="separate",
                 shift_coords=None, jitter_coords=None, rescale_coords=None):
        super(RopePositionEmbedding, self).__init__()
        assert normalize_coords in ["min", "max", "separate"]
        assert embedding_planes % (4 * head_nums) == 0
        both_periods = min_period is not None and max_period is not None
        if (base is None and not both_periods) or (base is not None and both_periods):
            raise ValueError("Either `base` or `min_period`+`max_period` must be provided.")
        head_planes = embedding_planes // head_nums
        self.min_period, self.max_period = min_period, max_period
        self.normalize_coords = normalize_coords
        self.shift_coords, self.jitter_coords, self.rescale_coords = shift_coords, jitter_coords, rescale_coords
        self.register_buffer("periods", torch.empty(head_planes // 4), persistent=True)
        if base is not None:
            periods = base ** (2 * torch.arange(head_planes // 4) / (head_planes // 2))
        else:
            base = self.max_period / self.min_period
            exponents = torch.linspace(0, 1, head_planes // 4)
            periods = base ** exponents
            periods = periods / base
            periods = periods * self.max_period
        self.periods.data = periods

    def forward(self, H, W):
        device = self.periods.device
        if self.normalize_coords == "max":
            coords_h, coords_w = torch.arange(0.5, H) / max(H, W), torch.arange(0.5, W) / max(H, W)
        elif self.normalize_coords == "min":
            coords_h, coords_w = torch.arange(0.5, H) / min(H, W), torch.arange(0.5, W) / min(H, W)
        else:
            coords_h, coords_w = torch.arange(0.5, H) / H, torch.arange(0.5, W) / W
        coords = torch.stack(torch.meshgrid(coords_h.to(device), coords_w.to(device), indexing="ij"), dim=-1).flatten(0, 1)
        coords = 2.0 * coords - 1.0
        if self.training and self.shift_coords is not None:
            shift_hw = torch.empty(2).uniform_(-self.shift_coords, self.shift_coords)
            coords = coords + shift_hw[None, :].to(device)
        if self.training and self.jitter_coords is not None:
            jitter_max = np.log(self.jitter_coords)
            jitter_hw = torch.empty(2).uniform_(-jitter_max, jitter_max).exp()
            coords = coords * jitter_hw[None, :].to(device)
        if self.training and self.rescale_coords is not None:
            rescale_max = np.log(self.rescale_coords)
            rescale_hw = torch.empty(1).uniform_(-rescale_max, rescale_max).exp()
            coords = coords * rescale_hw.to(device)
        angles = 2 * math.pi * coords[:, :, None] / self.periods[None, None, :]
        angles = angles.flatten(1, 2).tile(2)
        return torch.sin(angles), torch.cos(angles)


class LinearKMaskedBias(nn.Linear):
    """qkv projection whose k third carries no bias (reference :280-295: bias * bias_mask)"""

    def __init__(self, *args, **kwargs):
        super(LinearKMaskedBias, self).__init__(*args, **kwargs)
        o = self.out_features
        assert o % 3 == 0
        if self.bias is not None:
            self.register_buffer("bias_mask", torch.full_like(self.bias, fill_value=1))
            self.bias_mask[o // 3:2 * o // 3].fill_(0)

    def forward(self, input):
        masked_bias = self.bias * self.bias_mask.to(self.bias.dtype) if self.bias is not None else None
        return ops_tfm.linear_nd(input, self.weight, masked_bias)


class SelfAttention(nn.Module):

    def __init__(self, inplanes, head_nums=8, qkv_bias=False, proj_bias=True, attn_drop=0.0, proj_drop=0.0):
        super(SelfAttention, self).__init__()
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError('dropout is 0 in every reference DINOv3 configuration')
        if inplanes // head_nums != 64:
            raise NotImplementedError(f'head dim {inplanes // head_nums}: the streaming attention kernels are built for 64 here '
                                      '(every published DINOv3 size except the 7B model)')
        self.head_nums = head_nums
        self.scale = (inplanes // head_nums) ** -0.5
        self.qkv = LinearKMaskedBias(inplanes, inplanes * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(inplanes, inplanes, bias=proj_bias)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, attn_bias=None, rope=None):
        assert attn_bias is None
        qkv = self.qkv(x)                                                     # [B, N, 3C] = (3, heads, D) per token
        if rope is not None:
            sin, cos = rope
            if sin.dim() != 2:
                raise NotImplementedError('per-sample / per-head rope tables')
            qkv = ops_tfm.rope(qkv, sin, cos, self.head_nums, qkv.shape[1] - sin.shape[-2])
        x = ops_tfm.attention(qkv, self.head_nums, self.scale)
        return ops_tfm.linear_nd(x, self.proj.weight, self.proj.bias)


class SelfAttentionBlock(nn.Module):

    def __init__(self, inplanes, head_nums, ffn_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, drop=0.0, attn_drop=0.0,
                 init_values=None, drop_path=0.0, ffn_layer=Mlp):
        super(SelfAttentionBlock, self).__init__()
        self.norm1 = nn.LayerNorm(inplanes, eps=1e-6)
        self.attn = SelfAttention(inplanes, head_nums=head_nums, qkv_bias=qkv_bias, proj_bias=proj_bias, attn_drop=attn_drop,
                                  proj_drop=drop)
        self.ls1 = LayerScale(inplanes=inplanes, init_values=init_values) if init_values else nn.Identity()
        self.norm2 = nn.LayerNorm(inplanes, eps=1e-6)
        self.mlp = ffn_layer(inplanes=inplanes, hidden_planes=int(inplanes * ffn_ratio), planes=inplanes, drop_prob=drop,
                             bias=ffn_bias)
        self.ls2 = LayerScale(inplanes=inplanes, init_values=init_values) if init_values else nn.Identity()
        self.sample_drop_ratio = drop_path

    def _attn_branch(self, x, rope):
        return self.attn(ops_tfm.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps), rope=rope)

    def _mlp_branch(self, x):
        return self.mlp(ops_tfm.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps))

    def forward(self, x, rope=None):
        g1 = self.ls1.gamma if isinstance(self.ls1, LayerScale) else None
        g2 = self.ls2.gamma if isinstance(self.ls2, LayerScale) else None
        if self.training and self.sample_drop_ratio > 0.0:
            # stochastic depth over a random SUBSET of the batch (reference :412-444): the branch runs on the kept samples only and
            # its output is added back scaled by b / kept; index plumbing by torch, the branches on the kernels
            b = x.shape[0]
            keep = max(int(b * (1 - self.sample_drop_ratio)), 1)
            factor = b / keep
            idx1 = torch.randperm(b, device=x.device)[:keep]
            r1 = self._attn_branch(x[idx1], rope)
            x = torch.index_add(x, dim=0, source=(_residual(None, r1, g1) if g1 is not None else r1).to(x.dtype), index=idx1, alpha=factor)
            idx2 = torch.randperm(b, device=x.device)[:keep]
            r2 = self._mlp_branch(x[idx2])
            return torch.index_add(x, dim=0, source=(_residual(None, r2, g2) if g2 is not None else r2).to(x.dtype), index=idx2, alpha=factor)
        x = _residual(x, self._attn_branch(x, rope), g1)
        return _residual(x, self._mlp_branch(x), g2)


class DinoVisionTransformer(nn.Module):

    def __init__(self, patch_size=16, inplanes=3, embedding_planes=768, pos_embed_rope_base=100.0, pos_embed_rope_min_period=None,
                 pos_embed_rope_max_period=None, pos_embed_rope_normalize_coords="separate", pos_embed_rope_shift_coords=None,
                 pos_embed_rope_jitter_coords=None, pos_embed_rope_rescale_coords=None, block_nums=12, head_nums=12, ffn_ratio=4.0,
                 qkv_bias=True, drop_path_rate=0., layerscale_init=1e-5, ffn_layer="mlp", ffn_bias=True, proj_bias=True,
                 use_gradient_checkpoint=False):
        super(DinoVisionTransformer, self).__init__()
        assert pos_embed_rope_normalize_coords in ["min", "max", "separate"]
        self.patch_size = patch_size
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.patch_embed = PatchEmbed(inplanes=inplanes, planes=embedding_planes, kernel_size=patch_size, stride=patch_size,
                                      padding=0, has_norm=False)
        self.rope_embed = RopePositionEmbedding(
            embedding_planes=embedding_planes, head_nums=head_nums, base=pos_embed_rope_base, min_period=pos_embed_rope_min_period,
            max_period=pos_embed_rope_max_period, normalize_coords=pos_embed_rope_normalize_coords,
            shift_coords=pos_embed_rope_shift_coords, jitter_coords=pos_embed_rope_jitter_coords,
            rescale_coords=pos_embed_rope_rescale_coords)
        ffn_layer_cls = {"mlp": Mlp, "swiglu": SwiGLUFFN, "swiglu64": partial(SwiGLUFFN, align_to=64)}[ffn_layer]
        self.blocks = nn.ModuleList([
            SelfAttentionBlock(inplanes=embedding_planes, head_nums=head_nums, ffn_ratio=ffn_ratio, qkv_bias=qkv_bias,
                               proj_bias=proj_bias, ffn_bias=ffn_bias, init_values=layerscale_init, drop_path=drop_path_rate,
                               ffn_layer=ffn_layer_cls) for _ in range(block_nums)])
        self.norm = nn.LayerNorm(embedding_planes, eps=1e-6)
        self.out_channels = embedding_planes
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        batch, _, origin_h, origin_w = x.shape
        x = self.patch_embed(x)                                               # [B, H, W, C]
        _, H, W, _ = x.shape
        x = x.flatten(1, 2)
        rope_sincos = self.rope_embed(H=H, W=W)
        for block in self.blocks:
            if self.use_gradient_checkpoint:
                x = checkpoint(block, x, rope_sincos, use_reentrant=False)
            else:
                x = block(x, rope_sincos)
        x = ops_tfm.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x.reshape(batch, origin_h // self.patch_size, origin_w // self.patch_size, -1).permute(0, 3, 1, 2).contiguous()


def _dinov3vitbackbone(patch_size, embedding_planes, pos_embed_rope_normalize_coords, pos_embed_rope_rescale_coords, block_nums,
                       head_nums, ffn_ratio, qkv_bias, ffn_layer, pretrained_path='', **kwargs):
    model = DinoVisionTransformer(patch_size=patch_size, embedding_planes=embedding_planes,
                                  pos_embed_rope_normalize_coords=pos_embed_rope_normalize_coords,
                                  pos_embed_rope_rescale_coords=pos_embed_rope_rescale_coords, block_nums=block_nums,
                                  head_nums=head_nums, ffn_ratio=ffn_ratio, qkv_bias=qkv_bias, ffn_layer=ffn_layer, **kwargs)
    if pretrained_path:
        load_state_dict(pretrained_path, model)
    else:
        print('no backbone pretrained model!')
    return model


def _factory(embedding_planes, block_nums, head_nums, ffn_ratio, qkv_bias, ffn_layer):
    def build(patch_size=16, pretrained_path='', **kwargs):
        return _dinov3vitbackbone(patch_size=patch_size, embedding_planes=embedding_planes,
                                  pos_embed_rope_normalize_coords="separate", pos_embed_rope_rescale_coords=2,
                                  block_nums=block_nums, head_nums=head_nums, ffn_ratio=ffn_ratio, qkv_bias=qkv_bias,
                                  ffn_layer=ffn_layer, pretrained_path=pretrained_path, **kwargs)
    return build


# (embedding planes, blocks, heads, ffn ratio, qkv bias, ffn form): reference :575-721
dinov3_vit_small_patch16_backbone = _factory(384, 12, 6, 4, True, "mlp")
dinov3_vit_small_plus_patch16_backbone = _factory(384, 12, 6, 6, True, "swiglu")
dinov3_vit_base_patch16_backbone = _factory(768, 12, 12, 4, True, "mlp")
dinov3_vit_large_patch16_backbone = _factory(1024, 24, 16, 4, True, "mlp")
dinov3_vit_large_plus_patch16_backbone = _factory(1024, 24, 16, 6, True, "swiglu")
dinov3_vit_huge_plus_patch16_backbone = _factory(1280, 32, 20, 6, True, "swiglu")
dinov3_vit_7b_patch16_backbone = _factory(4096, 40, 32, 3, False, "swiglu64")        # head dim 128: SelfAttention refuses it
