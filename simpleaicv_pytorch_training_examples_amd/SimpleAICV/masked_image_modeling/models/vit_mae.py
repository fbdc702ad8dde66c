"""MAE pre-training model (masked autoencoder over ViT) on the MI355X HIP kernels -- SURVEY.md 8f rank 2: it is built
from the classification backbone's PatchEmbeddingBlock / TransformerEncoderLayer (reference
SimpleAICV/masked_image_modeling/models/vit_mae.py:16), so every block runs as the fused LayerNorm -> GEMM -> attention
-> GEMM(+residual) nodes of ops_tfm.

Interface contract (reference vit_mae.py): VITMAEPretrainModelEncoder (:25), VITMAEPretrainModelDecoder (:217),
VITMAEPretrainModel (:371; forward(x) -> (pred [B, L, p*p*3], mask [B, L]), images_to_patch, patch_to_images),
factories vit_{base_patch16,large_patch16,huge_patch14}_224_mae_pretrain_model (:469-515); identical constructor
arguments, parameter names / registration order (`encoder.{patch_embed.proj,cls_token,pos_embed,blocks.N.*,norm}`,
`decoder.{mask_token,pos_embed,blocks.N.*,norm,fc}`, `encoder_to_decoder`), frozen 2-d sin-cos position tables and the
same initialisation draw order (xavier on the flattened patch projection, N(0, 0.02) cls / mask tokens, xavier Linear).
The decoder of the base model has 16 heads of 32 channels: attention runs on the streaming kernels (head dim 32).

Token bookkeeping (random shuffle, gather of the kept quarter, un-shuffle with mask tokens) is a handful of [B, L]
index operations and [B, L, C] gathers per step; they stay tensor ops.  `noise` lets a caller supply the uniform
draws (the parity test replays the reference's CPU draws; the default draws on the device)."""
import numpy as np
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .... import ops_tfm
from ...classification.backbones.vit import PatchEmbeddingBlock, TransformerEncoderLayer

__all__ = [
    'vit_base_patch16_224_mae_pretrain_model',
    'vit_large_patch16_224_mae_pretrain_model',
    'vit_huge_patch14_224_mae_pretrain_model',
]


def sincos_position_table(planes, patch_nums, cls_token=True):
    """[1 + n*n, planes] fixed 2-d sin-cos table (reference :99-157): first half of the channels encodes the column
    index (the reference's meshgrid puts w first), second half the row index; each half is [sin | cos] over
    planes / 4 frequencies 1 / 10000^(i / (planes / 4)); a zero row in front for the class token."""
    assert planes % 2 == 0
    coords = np.arange(patch_nums, dtype=np.float32)
    grid = np.stack(np.meshgrid(coords, coords), axis=0).reshape(2, -1)          # [2, n*n]: (w index, h index)
    omega = np.arange(planes // 4, dtype=np.float32) / (planes / 4.)
    omega = 1. / 10000 ** omega
    halves = []
    for g in grid:
        out = np.einsum('m,d->md', g, omega)
        halves.append(np.concatenate([np.sin(out), np.cos(out)], axis=1))
    table = np.concatenate(halves, axis=1)
    if cls_token:
        table = np.concatenate([np.zeros([1, planes]), table], axis=0)
    return table


def _init_linear_and_norm(module):
    for m in module.modules():
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


def _run_blocks(blocks, x, use_checkpoint):
    for block in blocks:
        x = checkpoint(block, x, use_reentrant=False) if use_checkpoint else block(x)
    return x


class VITMAEPretrainModelEncoder(nn.Module):

    def __init__(self, patch_size, image_size, embedding_planes, block_nums, head_nums, feedforward_ratio,
                 mask_ratio=0.75, dropout_prob=0., use_gradient_checkpoint=False):
        super(VITMAEPretrainModelEncoder, self).__init__()
        self.image_size, self.patch_size, self.embedding_planes = image_size, patch_size, embedding_planes
        self.block_nums, self.head_nums, self.feedforward_ratio = block_nums, head_nums, feedforward_ratio
        self.mask_ratio = mask_ratio
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.patch_embed = PatchEmbeddingBlock(3, embedding_planes, kernel_size=patch_size, stride=patch_size, padding=0,
                                               groups=1, has_norm=False)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embedding_planes))
        self.pos_embed = nn.Parameter(torch.zeros(1, (image_size // patch_size) ** 2 + 1, embedding_planes),
                                      requires_grad=False)
        self.blocks = nn.ModuleList([
            TransformerEncoderLayer(embedding_planes, head_nums, feedforward_ratio=feedforward_ratio,
                                    dropout_prob=dropout_prob, drop_path_prob=0.) for _ in range(block_nums)])
        self.norm = nn.LayerNorm(embedding_planes, eps=1e-6)
        self.pos_embed.data.copy_(torch.from_numpy(
            sincos_position_table(embedding_planes, image_size // patch_size)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))                 # like nn.Linear, not like nn.Conv2d
        nn.init.normal_(self.cls_token, std=.02)
        _init_linear_and_norm(self)

    def random_masking(self, x, noise=None):
        """per-sample shuffle by argsort of uniform noise -> (keep_ids [B, keep], mask [B, N] (1 = removed),
        restore_ids [B, N])"""
        b, n, _ = x.shape
        keep_length = int(n * (1 - self.mask_ratio))
        if noise is None:
            noise = torch.rand(b, n, device=x.device)
        shuffle_ids = torch.argsort(noise.to(x.device), dim=1)
        restore_ids = torch.argsort(shuffle_ids, dim=1)
        mask = torch.ones([b, n], device=x.device)
        mask[:, :keep_length] = 0
        return shuffle_ids[:, :keep_length], torch.gather(mask, dim=1, index=restore_ids), restore_ids

    def forward(self, x, noise=None):
        x = self.patch_embed(x)                                                 # [B, L, C], compute dtype
        x = x + self.pos_embed[:, 1:, :].to(x.dtype)
        keep_ids, mask, restore_ids = self.random_masking(x, noise)
        x = torch.gather(x, dim=1, index=keep_ids.unsqueeze(-1).expand(-1, -1, x.shape[-1]))
        cls_token = (self.cls_token + self.pos_embed[:, :1, :]).to(x.dtype)
        x = torch.cat((cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        x = _run_blocks(self.blocks, x, self.use_gradient_checkpoint)
        x = ops_tfm.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x, mask, restore_ids


class VITMAEPretrainModelDecoder(nn.Module):

    def __init__(self, patch_size, image_size, embedding_planes, block_nums, head_nums, feedforward_ratio,
                 dropout_prob=0.1, use_gradient_checkpoint=False):
        super(VITMAEPretrainModelDecoder, self).__init__()
        self.patch_size, self.image_size, self.embedding_planes = patch_size, image_size, embedding_planes
        self.block_nums, self.head_nums, self.feedforward_ratio = block_nums, head_nums, feedforward_ratio
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embedding_planes))
        self.pos_embed = nn.Parameter(torch.zeros(1, (image_size // patch_size) ** 2 + 1, embedding_planes),
                                      requires_grad=False)
        self.blocks = nn.ModuleList([
            TransformerEncoderLayer(embedding_planes, head_nums, feedforward_ratio=feedforward_ratio,
                                    dropout_prob=dropout_prob, drop_path_prob=0.) for _ in range(block_nums)])
        self.norm = nn.LayerNorm(embedding_planes, eps=1e-6)
        self.fc = nn.Linear(embedding_planes, patch_size * patch_size * 3)
        self.pos_embed.data.copy_(torch.from_numpy(
            sincos_position_table(embedding_planes, image_size // patch_size)).float().unsqueeze(0))
        nn.init.normal_(self.mask_token, std=.02)
        _init_linear_and_norm(self)

    def forward(self, x, restore_ids):
        b, kept, c = x.shape
        mask_tokens = self.mask_token.to(x.dtype).expand(b, restore_ids.shape[1] + 1 - kept, -1)
        x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
        x_ = torch.gather(x_, dim=1, index=restore_ids.unsqueeze(-1).expand(-1, -1, c))       # un-shuffle
        x = torch.cat([x[:, :1, :], x_], dim=1) + self.pos_embed.to(x.dtype)
        x = _run_blocks(self.blocks, x, self.use_gradient_checkpoint)
        x = ops_tfm.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        x = ops_tfm.linear_nd(x, self.fc.weight, self.fc.bias)
        return x[:, 1:, :]


class VITMAEPretrainModel(nn.Module):

    def __init__(self, patch_size=16, image_size=224, mask_ratio=0.75, encoder_embedding_planes=768,
                 encoder_block_nums=12, encoder_head_nums=12, encoder_feedforward_ratio=4, encoder_dropout_prob=0.,
                 decoder_embedding_planes=384, decoder_block_nums=4, decoder_head_nums=6, decoder_feedforward_ratio=4,
                 decoder_dropout_prob=0., use_gradient_checkpoint=False):
        super(VITMAEPretrainModel, self).__init__()
        self.patch_size, self.image_size = patch_size, image_size
        assert self.image_size % self.patch_size == 0
        self.encoder = VITMAEPretrainModelEncoder(
            patch_size=patch_size, image_size=image_size, embedding_planes=encoder_embedding_planes,
            block_nums=encoder_block_nums, head_nums=encoder_head_nums, feedforward_ratio=encoder_feedforward_ratio,
            mask_ratio=mask_ratio, dropout_prob=encoder_dropout_prob, use_gradient_checkpoint=use_gradient_checkpoint)
        self.decoder = VITMAEPretrainModelDecoder(
            patch_size=patch_size, image_size=image_size, embedding_planes=decoder_embedding_planes,
            block_nums=decoder_block_nums, head_nums=decoder_head_nums, feedforward_ratio=decoder_feedforward_ratio,
            dropout_prob=decoder_dropout_prob, use_gradient_checkpoint=use_gradient_checkpoint)
        self.encoder_to_decoder = nn.Linear(encoder_embedding_planes, decoder_embedding_planes)
        _init_linear_and_norm(self.encoder_to_decoder)

    def forward(self, x, noise=None):
        x, mask, restore_ids = self.encoder(x, noise)
        x = ops_tfm.linear_nd(x, self.encoder_to_decoder.weight, self.encoder_to_decoder.bias)
        return self.decoder(x, restore_ids), mask

    def images_to_patch(self, images):
        """[N, 3, H, W] -> [N, L, p*p*3]: the regression target layout (h, w) x (p, q, c)"""
        n, p = self.image_size // self.patch_size, self.patch_size
        x = images.reshape(images.shape[0], 3, n, p, n, p)
        return torch.einsum('nchpwq->nhwpqc', x).reshape(images.shape[0], n * n, p * p * 3)

    def patch_to_images(self, x):
        h = int(x.shape[1] ** 0.5)
        p = self.patch_size
        images = x.reshape(x.shape[0], h, h, p, p, 3)
        return torch.einsum('nhwpqc->nchpwq', images).reshape(x.shape[0], 3, h * p, h * p)


def _vitmaepretrainmodel(**kwargs):
    return VITMAEPretrainModel(**kwargs)


def vit_base_patch16_224_mae_pretrain_model(**kwargs):
    return _vitmaepretrainmodel(patch_size=16, image_size=224, encoder_embedding_planes=768, encoder_block_nums=12,
                                encoder_head_nums=12, encoder_feedforward_ratio=4, encoder_dropout_prob=0.,
                                decoder_embedding_planes=512, decoder_block_nums=8, decoder_head_nums=16,
                                decoder_feedforward_ratio=4, decoder_dropout_prob=0., **kwargs)


def vit_large_patch16_224_mae_pretrain_model(**kwargs):
    return _vitmaepretrainmodel(patch_size=16, image_size=224, encoder_embedding_planes=1024, encoder_block_nums=24,
                                encoder_head_nums=16, encoder_feedforward_ratio=4, encoder_dropout_prob=0.,
                                decoder_embedding_planes=512, decoder_block_nums=8, decoder_head_nums=16,
                                decoder_feedforward_ratio=4, decoder_dropout_prob=0., **kwargs)


def vit_huge_patch14_224_mae_pretrain_model(**kwargs):
    return _vitmaepretrainmodel(patch_size=14, image_size=224, encoder_embedding_planes=1280, encoder_block_nums=32,
                                encoder_head_nums=16, encoder_feedforward_ratio=4, encoder_dropout_prob=0.,
                                decoder_embedding_planes=512, decoder_block_nums=8, decoder_head_nums=16,
                                decoder_feedforward_ratio=4, decoder_dropout_prob=0., **kwargs)
