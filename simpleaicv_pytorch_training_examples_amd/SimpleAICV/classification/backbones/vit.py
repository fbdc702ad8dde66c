"""ViT family on the MI355X HIP kernels -- drop-in for the reference factories.

Interface contract (reference SimpleAICV/classification/backbones/vit.py): class names
PatchEmbeddingBlock (:18), MultiHeadAttention (:50), FeedForward (:83), DropPathBlock (:102),
TransformerEncoderLayer (:138), ViT (:166), factories vit_base_patch16 / vit_large_patch16 /
vit_huge_patch14 (:273-282); identical constructor arguments, parameter names / shapes /
registration order (`patch_embed.proj.weight`, `cls_token`, `pos_embed`,
`blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*`, `norm.*`, `fc.*`) and the same
initialisation draw order (:228-237), so seeds and checkpoints carry over.

Execution: every pre-LN sub-layer is one fused autograd node (ops_tfm.AttnSubLayerFn /
MlpSubLayerFn): LayerNorm -> GEMM (bias epilogue) -> fused attention straight from the packed
qkv -> GEMM whose epilogue adds the residual and applies the per-sample drop-path factor.
LayerNorm eps is 1e-6 and the logits are scaled after q k^T (:72), as in the reference.
Head dim must be 64 (ViT-B/L), sequence <= 256 tokens.
"""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .... import ops, ops_tfm

__all__ = [
    'vit_base_patch16',
    'vit_large_patch16',
    'vit_huge_patch14',
]


class PatchEmbeddingBlock(nn.Module):

    def __init__(self, inplanes, planes, kernel_size, stride, padding, groups=1, has_norm=False):
        super(PatchEmbeddingBlock, self).__init__()
        if groups != 1 or padding != 0:
            raise NotImplementedError('patch embedding is a plain stride-p convolution on the hot path')
        bias = False if has_norm else True
        self.proj = nn.Conv2d(inplanes, planes, kernel_size, stride=stride, padding=padding, groups=groups, bias=bias)
        self.norm = nn.LayerNorm(inplanes, eps=1e-6) if has_norm else nn.Identity()
        self.stride = stride
        self.has_norm = has_norm

    def forward(self, x):
        x = ops_tfm.patch_embed(x, self.proj.weight, self.proj.bias, self.stride)     # [B, N, C]
        if self.has_norm:
            x = ops_tfm.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x


class MultiHeadAttention(nn.Module):

    def __init__(self, inplanes, head_nums=8, dropout_prob=0.):
        super(MultiHeadAttention, self).__init__()
        if dropout_prob != 0.:
            raise NotImplementedError('attention dropout is 0 in every reference ViT config')
        self.head_nums = head_nums
        self.scale = (inplanes // head_nums) ** -0.5
        self.qkv = nn.Linear(inplanes, inplanes * 3)
        self.proj = nn.Linear(inplanes, inplanes)
        self.dropout = nn.Dropout(dropout_prob)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        qkv = ops_tfm.linear_nd(x, self.qkv.weight, self.qkv.bias)
        x = ops_tfm.attention(qkv, self.head_nums, self.scale)
        return ops_tfm.linear_nd(x, self.proj.weight, self.proj.bias)


class FeedForward(nn.Module):

    def __init__(self, inplanes, feedforward_planes, dropout_prob=0.):
        super(FeedForward, self).__init__()
        if dropout_prob != 0.:
            raise NotImplementedError('MLP dropout is 0 in every reference ViT config')
        self.fc1 = nn.Linear(inplanes, feedforward_planes)
        self.gelu = nn.GELU()
        self.fc2 = nn.Linear(feedforward_planes, inplanes)
        self.drop = nn.Dropout(dropout_prob)

    def forward(self, x):
        x = ops_tfm.gelu(ops_tfm.linear_nd(x, self.fc1.weight, self.fc1.bias))
        return ops_tfm.linear_nd(x, self.fc2.weight, self.fc2.bias)


class DropPathBlock(nn.Module):
    """Stochastic depth per sample.  `sample_scale` draws the per-sample factor (0 or 1/keep) that
    the fused sub-layer applies in its GEMM epilogue; calling the module directly multiplies."""

    def __init__(self, drop_path_prob=0., scale_by_keep=True):
        super(DropPathBlock, self).__init__()
        assert drop_path_prob >= 0.
        self.drop_path_prob = drop_path_prob
        self.keep_path_prob = 1 - drop_path_prob
        self.scale_by_keep = scale_by_keep

    def sample_scale(self, batch, device):
        if self.drop_path_prob == 0. or not self.training:
            return None
        w = torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(self.keep_path_prob)
        if self.keep_path_prob > 0. and self.scale_by_keep:
            w.div_(self.keep_path_prob)
        return w

    def forward(self, x):
        w = self.sample_scale(x.shape[0], x.device)
        if w is None:
            return x
        return w.view((x.shape[0],) + (1,) * (x.dim() - 1)).to(x.dtype) * x


class TransformerEncoderLayer(nn.Module):

    def __init__(self, inplanes, head_nums, feedforward_ratio=4, dropout_prob=0., drop_path_prob=0.):
        super(TransformerEncoderLayer, self).__init__()
        self.norm1 = nn.LayerNorm(inplanes, eps=1e-6)
        self.attn = MultiHeadAttention(inplanes, head_nums, dropout_prob=dropout_prob)
        self.norm2 = nn.LayerNorm(inplanes, eps=1e-6)
        self.mlp = FeedForward(inplanes, int(inplanes * feedforward_ratio), dropout_prob=dropout_prob)
        # if test model,drop_path must set to 0.
        self.drop_path = DropPathBlock(drop_path_prob) if drop_path_prob > 0. else nn.Identity()

    def _scale(self, x):
        if isinstance(self.drop_path, DropPathBlock):
            return self.drop_path.sample_scale(x.shape[0], x.device)
        return None

    def forward(self, x, scales=None):
        """scales: the two per-sample drop-path factors of this layer, drawn by the caller for all layers at once (ViT.forward);
        None = draw them here (one bernoulli_ + one div_ per sub-layer)."""
        s1, s2 = scales if scales is not None else (self._scale(x), self._scale(x))
        x = ops_tfm.attn_sublayer(x, self.norm1, self.attn, s1)
        x = ops_tfm.mlp_sublayer(x, self.norm2, self.mlp, s2)
        return x


class ViT(nn.Module):

    def __init__(self, patch_size, embedding_planes, block_nums, head_nums, feedforward_ratio, image_size=224,
                 dropout_prob=0., drop_path_prob=0., global_pool=False, num_classes=1000,
                 use_gradient_checkpoint=False):
        super(ViT, self).__init__()
        self.image_size = image_size
        self.patch_size = patch_size
        self.embedding_planes = embedding_planes
        self.block_nums = block_nums
        self.head_nums = head_nums
        self.feedforward_ratio = feedforward_ratio
        self.global_pool = global_pool
        self.num_classes = num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        if dropout_prob != 0.:
            raise NotImplementedError('dropout_prob is 0 in every reference ViT config')

        self.patch_embed = PatchEmbeddingBlock(3, self.embedding_planes, kernel_size=self.patch_size,
                                               stride=self.patch_size, padding=0, groups=1, has_norm=False)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.embedding_planes))
        self.pos_embed = nn.Parameter(torch.ones(1, (self.image_size // self.patch_size) ** 2 + 1,
                                                 self.embedding_planes))
        self.embedding_dropout = nn.Dropout(dropout_prob)

        rates = [0. if drop_path_prob == 0. else drop_path_prob * (i / (self.block_nums - 1))
                 for i in range(self.block_nums)]
        self.blocks = nn.ModuleList([
            TransformerEncoderLayer(self.embedding_planes, self.head_nums, feedforward_ratio=self.feedforward_ratio,
                                    dropout_prob=dropout_prob, drop_path_prob=rates[i])
            for i in range(self.block_nums)])
        self.norm = nn.LayerNorm(self.embedding_planes, eps=1e-6)
        self.fc = nn.Linear(self.embedding_planes, self.num_classes)

        # same draw order as reference vit.py:228-237
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        nn.init.trunc_normal_(self.fc.weight, std=2e-5)
        nn.init.zeros_(self.fc.bias)

    def _drop_path_scales(self, batch, device):
        """All stochastic-depth factors of a forward pass from ONE uniform draw: row 2i / 2i+1 = the per-sample factor (0 or
        1 / keep) of layer i's attention / MLP sub-layer (reference vit.py:102-133 draws one bernoulli per DropPathBlock call;
        the masks are not RNG-identical to the reference either way).  Three launches instead of 4 per layer."""
        if not self.training or self.use_gradient_checkpoint:
            return None
        keeps = [blk.drop_path.keep_path_prob if isinstance(blk.drop_path, DropPathBlock) else 1.0 for blk in self.blocks]
        if all(k >= 1.0 for k in keeps):
            return None
        cache = getattr(self, '_keep_cache', None)
        if cache is None or cache[0].device != device:
            keep = torch.tensor([k for k in keeps for _ in (0, 1)], dtype=torch.float32, device=device).view(-1, 1)
            scale_by = torch.tensor([(1.0 / k if (k > 0. and blk.drop_path.scale_by_keep) else 1.0) if isinstance(blk.drop_path, DropPathBlock) else 1.0
                                     for k, blk in zip(keeps, self.blocks) for _ in (0, 1)], dtype=torch.float32, device=device).view(-1, 1)
            cache = self._keep_cache = (keep, scale_by)
        keep, scale_by = cache
        u = torch.rand(keep.shape[0], batch, dtype=torch.float32, device=device)
        return (u < keep).to(torch.float32).mul_(scale_by)

    def forward(self, x):
        x = self.patch_embed(x)                                             # [B, N, C], compute dtype
        # token assembly (cls concat + position embedding): two small elementwise ops on [B, N+1, C]
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1).to(x.dtype), x), dim=1)
        x = x + self.pos_embed.to(x.dtype)
        scales = self._drop_path_scales(x.shape[0], x.device)
        for i, block in enumerate(self.blocks):
            if self.use_gradient_checkpoint:
                x = checkpoint(block, x, use_reentrant=False)
            else:
                dropped = scales is not None and isinstance(block.drop_path, DropPathBlock) and block.drop_path.drop_path_prob > 0.
                x = block(x, (scales[2 * i], scales[2 * i + 1]) if dropped else (None, None) if scales is not None else None)
        if self.global_pool:
            x = x[:, 1:, :].float().mean(dim=1).to(x.dtype)                # global pool without cls token
            x = ops_tfm.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            x = ops_tfm.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)[:, 0]
        return ops.linear(x.contiguous(), self.fc.weight, self.fc.bias, out_f32=True)


def _vit(patch_size, embedding_planes, block_nums, head_nums, feedforward_ratio, **kwargs):
    return ViT(patch_size, embedding_planes, block_nums, head_nums, feedforward_ratio, **kwargs)


def vit_base_patch16(**kwargs):
    return _vit(16, 768, 12, 12, 4, **kwargs)


def vit_large_patch16(**kwargs):
    return _vit(16, 1024, 24, 16, 4, **kwargs)


def vit_huge_patch14(**kwargs):
    return _vit(14, 1280, 32, 16, 4, **kwargs)
