"""SAM ViT image encoder on the MI355X HIP kernels -- drop-in for the reference module.

Interface contract (reference SimpleAICV/interactive_segmentation/models/segment_anything/
image_encoder.py): PatchEmbed (:8), Attention (:147), MLPBlock (:187), Block (:201), LayerNorm2d (:242),
ViTImageEncoder (:259) with identical constructor arguments and parameter names / shapes
(`patch_embed.proj.*`, `pos_embed`, `blocks.N.{norm1,attn.qkv,attn.proj,attn.rel_pos_h,attn.rel_pos_w,
norm2,mlp.lin1,mlp.lin2}.*`, `neck.{0,2}.weight`, `neck.{1,3}.{weight,bias}`), so reference checkpoints load.

Execution: tokens stay [B, H, W, C] (the NHWC conv output is the token grid); each Block is two fused
autograd nodes (ops_tfm.SamAttnSubLayerFn / MlpSubLayerFn).  Attention streams keys through LDS with
an online softmax and adds the decomposed relative-position logits in registers: the 4096 x 4096
matrix of the global blocks (403 MB per image per block in the reference) never exists in HBM.
Windows are padded with zeros AFTER norm1, and the padded tokens take part in attention as keys
(their q/k/v are the qkv bias), exactly as the reference does.  The neck runs on NHWC: the 1x1 and
3x3 bias-free convs are implicit GEMMs and LayerNorm2d is the row LayerNorm over the channel axis.
Head dim must be 64 (ViT-B / ViT-L encoders).
"""
import os

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ..... import ops, ops_tfm


class PatchEmbed(nn.Module):

    def __init__(self, inplanes=3, planes=768, kernel_size=16, stride=16, padding=0):
        super(PatchEmbed, self).__init__()
        if padding != 0:
            raise NotImplementedError('patch embedding is a plain stride-p convolution on the hot path')
        self.proj = nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=stride, padding=padding)
        self.stride = stride

    def forward(self, x):
        b, _, h, w = x.shape
        t = ops_tfm.patch_embed(x, self.proj.weight, self.proj.bias, self.stride)       # [B, N, C]
        return t.view(b, h // self.stride, w // self.stride, -1)                        # B H W C


class Attention(nn.Module):
    """Parameter holder; the computation lives in ops_tfm.SamAttnSubLayerFn (called by Block)."""

    def __init__(self, inplanes, head_nums=8, input_size=None):
        super(Attention, self).__init__()
        self.head_nums = head_nums
        head_planes = inplanes // head_nums
        if head_planes != 64:
            raise NotImplementedError(f'head dim {head_planes}: the streaming attention kernel of the SAM '
                                      'encoder is instantiated for 64')
        self.scale = head_planes ** -0.5
        self.qkv = nn.Linear(inplanes, inplanes * 3)
        self.proj = nn.Linear(inplanes, inplanes)
        assert input_size is not None, "Input size must be provided if using relative positional encoding."
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_planes))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_planes))


class MLPBlock(nn.Module):

    def __init__(self, inplanes, mlp_planes):
        super(MLPBlock, self).__init__()
        self.lin1 = nn.Linear(inplanes, mlp_planes)
        self.lin2 = nn.Linear(mlp_planes, inplanes)
        self.act = nn.GELU()

    def forward(self, x):
        x = ops_tfm.gelu(ops_tfm.linear_nd(x, self.lin1.weight, self.lin1.bias))
        return ops_tfm.linear_nd(x, self.lin2.weight, self.lin2.bias)


class Block(nn.Module):

    def __init__(self, inplanes, head_nums, mlp_ratio=4.0, input_size=None, window_size=0):
        super(Block, self).__init__()
        self.norm1 = nn.LayerNorm(inplanes, eps=1e-6)
        self.attn = Attention(inplanes=inplanes, head_nums=head_nums,
                              input_size=input_size if window_size == 0 else (window_size, window_size))
        self.norm2 = nn.LayerNorm(inplanes, eps=1e-6)
        self.mlp = MLPBlock(inplanes=inplanes, mlp_planes=int(inplanes * mlp_ratio))
        self.window_size = window_size

    def forward(self, x):
        b, h, w, c = x.shape
        x = ops_tfm.sam_attn_sublayer(x, self.norm1, self.attn, self.window_size)
        m = self.mlp
        x = ops_tfm.MlpSubLayerFn.apply(x.view(b, h * w, c), self.norm2.weight, self.norm2.bias, m.lin1.weight,
                                        m.lin1.bias, m.lin2.weight, m.lin2.bias, None, self.norm2.eps)
        return x.view(b, h, w, c)


class LayerNorm2d(nn.Module):
    """LayerNorm over the channel axis of an NCHW-shaped tensor (biased variance, eps inside the sqrt)."""

    def __init__(self, inplanes, eps=1e-6):
        super(LayerNorm2d, self).__init__()
        self.weight = nn.Parameter(torch.ones(inplanes))
        self.bias = nn.Parameter(torch.zeros(inplanes))
        self.eps = eps

    def forward(self, x):
        t = x.permute(0, 2, 3, 1)                       # NHWC view of a channels-last tensor: no copy
        t = ops_tfm.layer_norm(t, self.weight, self.bias, self.eps)
        return t.permute(0, 3, 1, 2)


class _NeckConv(nn.Conv2d):
    """nn.Conv2d parameter layout, implicit-GEMM execution."""

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0])


class ViTImageEncoder(nn.Module):

    def __init__(self, image_size=1024, patch_size=16, inplanes=3, embedding_planes=768, block_nums=12,
                 head_nums=12, mlp_ratio=4, out_planes=256, window_size=0, global_attn_indexes=(),
                 use_gradient_checkpoint=False):
        super(ViTImageEncoder, self).__init__()
        self.image_size = image_size
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.patch_embed = PatchEmbed(inplanes=inplanes, planes=embedding_planes, kernel_size=patch_size,
                                      stride=patch_size, padding=0)
        self.pos_embed = nn.Parameter(
            torch.zeros(1, image_size // patch_size, image_size // patch_size, embedding_planes))
        blocks = []
        for i in range(block_nums):
            blocks.append(Block(inplanes=embedding_planes, head_nums=head_nums, mlp_ratio=mlp_ratio,
                                input_size=(image_size // patch_size, image_size // patch_size),
                                window_size=window_size if i not in global_attn_indexes else 0))
        self.blocks = nn.ModuleList(blocks)
        self.neck = nn.Sequential(
            _NeckConv(embedding_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=False),
            LayerNorm2d(out_planes),
            _NeckConv(out_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=False),
            LayerNorm2d(out_planes))

    def _recompute(self, x):
        """`use_gradient_checkpoint=True` (reference sam_b_training/train_config.py:21) asks for per-block recomputation
        to fit a 24-80 GB GPU; on MI355X it is HONOURED ONLY WHEN THE ACTIVATIONS WOULD NOT FIT: the blocks keep about
        34 bytes per token and channel for backward (x, LN output, qkv, attention output, the two MLP tensors, lse),
        1.5 GB per 1024 x 1024 image for ViT-B -- per-GPU batch 20 is 31 GB of 288 GB -- and recomputation costs a
        third more encoder time (measured: 167 vs 125 ms at batch 20).  SAICV_ACTIVATION_CHECKPOINT=1 / 0 forces it."""
        if not (self.use_gradient_checkpoint and torch.is_grad_enabled()):
            return False
        forced = os.environ.get('SAICV_ACTIVATION_CHECKPOINT')
        if forced is not None:
            return forced == '1'
        need = x.shape[0] * x.shape[1] * x.shape[2] * 1.2 * x.shape[3] * 34 * len(self.blocks)     # 1.2: window padding
        free = torch.cuda.mem_get_info(x.device)[0] if x.is_cuda else 0
        return need > 0.6 * free

    def forward(self, x):
        x = self.patch_embed(x)                                  # [B, H, W, C], compute dtype
        x = x + self.pos_embed.to(x.dtype)
        recompute = self._recompute(x)
        for block in self.blocks:
            x = checkpoint(block, x, use_reentrant=False) if recompute else block(x)
        x = x.permute(0, 3, 1, 2)                                # NCHW shape over NHWC memory
        if recompute:
            x = checkpoint(self.neck, x, use_reentrant=False)
        else:
            x = self.neck(x)
        return x
