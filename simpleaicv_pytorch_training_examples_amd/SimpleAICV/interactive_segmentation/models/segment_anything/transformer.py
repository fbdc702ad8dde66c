"""SAM two-way transformer on the MI355X HIP kernels -- drop-in for the reference module.

Interface contract (reference SimpleAICV/interactive_segmentation/models/segment_anything/transformer.py):
MLPBlock (:7), Attention (:21), TwoWayAttentionBlock (:70), TwoWayTransformer (:128); identical constructor
arguments and parameter names (`layers.N.{self_attn,cross_attn_token_to_image,cross_attn_image_to_token}.
{q,k,v,out}_proj.*`, `layers.N.norm{1..4}.*`, `layers.N.mlp.lin{1,2}.*`, `final_attn_token_to_image.*`,
`norm_final_attn.*`).

Execution: projections are implicit-GEMM linears, LayerNorms the wavefront-reduction kernel, and every
attention (7-11 prompt tokens <-> 4096 image tokens, both directions) goes through the streaming attention
kernel, so no [heads, 4096, tokens] logits tensor is materialised.  The cross attentions run at 128
channels / 8 heads = head dim 16; the kernel is instantiated for 32 and 64, so q / k / v heads are
zero-padded to 32 (exact: zero columns add nothing to q.k and produce zero output columns that are dropped).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..... import ops_tfm


class MLPBlock(nn.Module):

    def __init__(self, inplanes, mlp_planes):
        super(MLPBlock, self).__init__()
        self.lin1 = nn.Linear(inplanes, mlp_planes)
        self.lin2 = nn.Linear(mlp_planes, inplanes)
        self.act = nn.ReLU(inplace=True)

    def forward(self, x):
        x = torch.relu(ops_tfm.linear_nd(x, self.lin1.weight, self.lin1.bias))
        return ops_tfm.linear_nd(x, self.lin2.weight, self.lin2.bias)


def _pad_heads(x, head_nums, to):
    b, n, c = x.shape
    d = c // head_nums
    if d == to:
        return x
    return F.pad(x.view(b, n, head_nums, d), (0, to - d)).view(b, n, head_nums * to)


class Attention(nn.Module):

    def __init__(self, inplanes, head_nums, downsample_rate=1):
        super(Attention, self).__init__()
        inter_planes = inplanes // downsample_rate
        self.head_nums = head_nums
        assert inter_planes % head_nums == 0, "head_nums must divide inplanes."
        self.q_proj = nn.Linear(inplanes, inter_planes)
        self.k_proj = nn.Linear(inplanes, inter_planes)
        self.v_proj = nn.Linear(inplanes, inter_planes)
        self.out_proj = nn.Linear(inter_planes, inplanes)

    def forward(self, q, k, v):
        q = ops_tfm.linear_nd(q, self.q_proj.weight, self.q_proj.bias)
        k = ops_tfm.linear_nd(k, self.k_proj.weight, self.k_proj.bias)
        v = ops_tfm.linear_nd(v, self.v_proj.weight, self.v_proj.bias)
        c = q.shape[-1]
        d = c // self.head_nums
        if d > 64:
            raise NotImplementedError(f'head dim {d} > 64')
        dk = 32 if d <= 32 else 64
        out = ops_tfm.stream_attention(_pad_heads(q, self.head_nums, dk), _pad_heads(k, self.head_nums, dk),
                                       _pad_heads(v, self.head_nums, dk), self.head_nums, 1.0 / math.sqrt(d))
        if dk != d:
            b, n, _ = out.shape
            out = out.view(b, n, self.head_nums, dk)[..., :d].reshape(b, n, c)
        return ops_tfm.linear_nd(out, self.out_proj.weight, self.out_proj.bias)


def _ln(norm, x):
    return ops_tfm.layer_norm(x, norm.weight, norm.bias, norm.eps)


class TwoWayAttentionBlock(nn.Module):

    def __init__(self, inplanes, head_nums, mlp_planes=2048, attention_downsample_rate=2, skip_first_layer_pe=False):
        super(TwoWayAttentionBlock, self).__init__()
        self.self_attn = Attention(inplanes, head_nums)
        self.norm1 = nn.LayerNorm(inplanes)
        self.cross_attn_token_to_image = Attention(inplanes, head_nums, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(inplanes)
        self.mlp = MLPBlock(inplanes, mlp_planes)
        self.norm3 = nn.LayerNorm(inplanes)
        self.norm4 = nn.LayerNorm(inplanes)
        self.cross_attn_image_to_token = Attention(inplanes, head_nums, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe

    def forward(self, queries, keys, query_pe, key_pe):
        if self.skip_first_layer_pe:
            queries = self.self_attn(q=queries, k=queries, v=queries)
        else:
            q = queries + query_pe
            queries = queries + self.self_attn(q=q, k=q, v=queries)
        queries = _ln(self.norm1, queries)

        q = queries + query_pe
        k = keys + key_pe
        queries = queries + self.cross_attn_token_to_image(q=q, k=k, v=keys)
        queries = _ln(self.norm2, queries)

        queries = queries + self.mlp(queries)
        queries = _ln(self.norm3, queries)

        q = queries + query_pe
        k = keys + key_pe
        keys = keys + self.cross_attn_image_to_token(q=k, k=q, v=queries)
        keys = _ln(self.norm4, keys)
        return queries, keys


class TwoWayTransformer(nn.Module):

    def __init__(self, block_nums=2, embedding_planes=256, head_nums=8, mlp_planes=2048, attention_downsample_rate=2):
        super(TwoWayTransformer, self).__init__()
        self.layers = nn.ModuleList()
        for i in range(block_nums):
            self.layers.append(
                TwoWayAttentionBlock(inplanes=embedding_planes, head_nums=head_nums, mlp_planes=mlp_planes,
                                     attention_downsample_rate=attention_downsample_rate,
                                     skip_first_layer_pe=(i == 0)))
        self.final_attn_token_to_image = Attention(embedding_planes, head_nums,
                                                   downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_planes)

    def forward(self, image_embedding, image_pe, point_embedding):
        # B x C x H x W (NHWC memory from the encoder neck) -> B x HW x C: a view, not a copy
        image_embedding = image_embedding.flatten(2).permute(0, 2, 1)
        image_pe = image_pe.flatten(2).permute(0, 2, 1)
        dt = image_embedding.dtype
        queries = point_embedding.to(dt)
        point_embedding = queries
        keys = image_embedding
        image_pe = image_pe.to(dt)
        for layer in self.layers:
            queries, keys = layer(queries=queries, keys=keys, query_pe=point_embedding, key_pe=image_pe)
        q = queries + point_embedding
        k = keys + image_pe
        queries = queries + self.final_attn_token_to_image(q=q, k=k, v=keys)
        queries = _ln(self.norm_final_attn, queries)
        return queries, keys
