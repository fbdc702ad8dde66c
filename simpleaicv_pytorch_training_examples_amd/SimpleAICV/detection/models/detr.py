"""DETR on the MI355X HIP kernels -- drop-in for the reference factories.

Interface contract (reference SimpleAICV/detection/models/detr.py): ActivationBlock (:28), TransformerEncoderLayer
(:44), TransformerDecoderLayer (:93), DETRTransformer (:183), DETR (:273), factories resnet{18,34,50,101,152}_detr
(:383-410); identical constructor arguments, state_dict keys (the attention parameters live in
nn.MultiheadAttention holders: `attention.in_proj_weight [3C, C]`, `attention.out_proj.*`, ...) and init order.

Execution (tokens kept batch-first [B, L, C]; the reference's [L, B, C] is the same math):
  * q / k / v projections are slices of the packed in_proj weight fed to the implicit-GEMM linear (q and k of a
    self-attention share their input, src + pos, so they are ONE GEMM over the first 2C rows);
  * attention is the streaming kernel at head dim 32; DETR hands nn.MultiheadAttention a FLOAT key_padding_mask
    (`masks.float()`, :343), which PyTorch ADDS to the logits -- padded keys get +1.0, not -inf (SURVEY.md
    section 7) -- reproduced exactly through the kernel's per-key additive bias; attention-probability dropout
    (p = 0.1) is applied inside the kernel from a counter-based hash;
  * post-LN residual blocks: LayerNorm kernel; FFN = two linears with ReLU.
Residual / FFN dropouts act on [B, L, 256] activations and use the framework's dropout op.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .... import ops, ops_tfm
from . import backbones
from .backbones.detr_resnet import PositionEmbeddingBlock
from .head import DETRClsRegHead

__all__ = [
    'resnet18_detr',
    'resnet34_detr',
    'resnet50_detr',
    'resnet101_detr',
    'resnet152_detr',
]


class ActivationBlock(nn.Module):

    def __init__(self, act_type='relu'):
        super(ActivationBlock, self).__init__()
        assert act_type in ['relu', 'gelu'], 'Unsupport activation function!'
        self.act_type = act_type
        self.act = nn.ReLU(inplace=True) if act_type == 'relu' else nn.GELU()

    def forward(self, x):
        return torch.relu(x) if self.act_type == 'relu' else ops_tfm.gelu(x)


def _mha(mha, q_in, k_in, v_in, key_bias, same_qk):
    """nn.MultiheadAttention.forward(need_weights=False) with batch-first [B, L, C] tensors."""
    c = mha.embed_dim
    w, b = mha.in_proj_weight, mha.in_proj_bias
    # the packed parameters enter whole (row ranges): their gradients land in the arena rows directly
    # head dimension 64: the streaming attention kernels stage K and V with ONE row stride (csrc/attn_stream.hip), so K must not
    # be a column slice of the packed [q | k] projection (row stride 2c against V's c) -- project it on its own
    p = mha.dropout if mha.training else 0.0
    scale = (c // mha.num_heads) ** -0.5
    v = ops_tfm.linear_rows(v_in, w, b, 2 * c, 3 * c)
    if same_qk and c // mha.num_heads != 64:
        # one projection for [q | k]; the halves are taken inside the attention Function (r06: slicing here cost five autograd launches per
        # call on the way back)
        qk = ops_tfm.linear_rows(q_in, w, b, 0, 2 * c)
        out = ops_tfm.stream_attention_packed_qk(qk, v, mha.num_heads, scale, key_bias, p)
    else:
        q = ops_tfm.linear_rows(q_in, w, b, 0, c)
        k = ops_tfm.linear_rows(k_in, w, b, c, 2 * c)
        out = ops_tfm.stream_attention(q, k, v, mha.num_heads, scale, key_bias, p)
    return ops_tfm.linear_nd(out, mha.out_proj.weight, mha.out_proj.bias)


def _ln(norm, x):
    return ops_tfm.layer_norm(x, norm.weight, norm.bias, norm.eps)


def _res_ln(norm, x, branch, dropout):
    """norm(x + dropout(branch)) (reference detr.py:89,92,114,118,122): one fused kernel each way while the dropout is active on the GPU
    (r06: a DETR step spent 42 dropout + 42 masked-scale + ~60 add launches on these thirty residuals), the plain ops otherwise"""
    if dropout.training and dropout.p > 0. and x.is_cuda:
        return ops_tfm.dropout_add_layer_norm(x, branch, norm.weight, norm.bias, dropout.p, norm.eps)
    return _ln(norm, x + dropout(branch))


def _ffn_hidden(act, dropout, h):
    """dropout(activation(h)) of the feed-forward (reference detr.py:90-91, 120-121): one kernel each way for ReLU while the dropout is
    active on the GPU (r06), the plain ops otherwise"""
    if act.act_type == 'relu' and dropout.training and dropout.p > 0. and h.is_cuda and h.numel() % 8 == 0:
        return ops_tfm.relu_dropout(h, dropout.p)
    return dropout(act(h))


class TransformerEncoderLayer(nn.Module):

    def __init__(self, hidden_planes, head_nums, feedforward_ratio=4, dropout_prob=0.1, act_type="relu"):
        super(TransformerEncoderLayer, self).__init__()
        self.attention = nn.MultiheadAttention(hidden_planes, head_nums, dropout=dropout_prob)
        self.linear1 = nn.Linear(hidden_planes, int(hidden_planes * feedforward_ratio))
        self.linear2 = nn.Linear(int(hidden_planes * feedforward_ratio), hidden_planes)
        self.norm1 = nn.LayerNorm(hidden_planes)
        self.norm2 = nn.LayerNorm(hidden_planes)
        self.act = ActivationBlock(act_type)
        self.dropout = nn.Dropout(dropout_prob)

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos=None):
        """src / pos: [B, L, C]; src_key_padding_mask: fp32 [B, L] added to the logits."""
        assert src_mask is None
        qk = src + pos if pos is not None else src
        src2 = _mha(self.attention, qk, qk, src, src_key_padding_mask, True)
        src = _res_ln(self.norm1, src, src2, self.dropout)
        src2 = ops_tfm.linear_nd(_ffn_hidden(self.act, self.dropout, ops_tfm.linear_nd(src, self.linear1.weight, self.linear1.bias)),
                                 self.linear2.weight, self.linear2.bias)
        return _res_ln(self.norm2, src, src2, self.dropout)


class TransformerDecoderLayer(nn.Module):

    def __init__(self, hidden_planes, head_nums, feedforward_ratio=4, dropout_prob=0.1, act_type="relu"):
        super(TransformerDecoderLayer, self).__init__()
        self.attention = nn.MultiheadAttention(hidden_planes, head_nums, dropout=dropout_prob)
        self.multihead_attention = nn.MultiheadAttention(hidden_planes, head_nums, dropout=dropout_prob)
        self.linear1 = nn.Linear(hidden_planes, int(hidden_planes * feedforward_ratio))
        self.linear2 = nn.Linear(int(hidden_planes * feedforward_ratio), hidden_planes)
        self.norm1 = nn.LayerNorm(hidden_planes)
        self.norm2 = nn.LayerNorm(hidden_planes)
        self.norm3 = nn.LayerNorm(hidden_planes)
        self.activation = ActivationBlock(act_type)
        self.dropout = nn.Dropout(dropout_prob)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None, memory_pos=None):
        """memory_pos: `memory + pos` computed once by the caller (the six decoder layers add the same two tensors: reference
        detr.py:112 does it in every layer)"""
        assert tgt_mask is None and memory_mask is None
        qk = tgt + query_pos if query_pos is not None else tgt
        tgt2 = _mha(self.attention, qk, qk, tgt, tgt_key_padding_mask, True)
        tgt = _res_ln(self.norm1, tgt, tgt2, self.dropout)
        q = tgt + query_pos if query_pos is not None else tgt
        k = memory_pos if memory_pos is not None else (memory + pos if pos is not None else memory)
        tgt2 = _mha(self.multihead_attention, q, k, memory, memory_key_padding_mask, False)
        tgt = _res_ln(self.norm2, tgt, tgt2, self.dropout)
        tgt2 = ops_tfm.linear_nd(
            _ffn_hidden(self.activation, self.dropout, ops_tfm.linear_nd(tgt, self.linear1.weight, self.linear1.bias)),
            self.linear2.weight, self.linear2.bias)
        return _res_ln(self.norm3, tgt, tgt2, self.dropout)


class DETRTransformer(nn.Module):

    def __init__(self, inplanes=256, head_nums=8, feedforward_ratio=4, encoder_layer_nums=6, decoder_layer_nums=6,
                 dropout_prob=0.1, act_type='relu'):
        super(DETRTransformer, self).__init__()
        self.inplanes = inplanes
        self.head_nums = head_nums
        self.feedforward_ratio = feedforward_ratio
        self.encoder_layer_nums = encoder_layer_nums
        self.decoder_layer_nums = decoder_layer_nums
        self.dropout_prob = dropout_prob
        self.act_type = act_type
        if inplanes // head_nums not in (32, 64):
            raise NotImplementedError(f'head dim {inplanes // head_nums}: the attention kernel is instantiated for 32 / 64')
        self.encoder_blocks = nn.ModuleList([
            TransformerEncoderLayer(self.inplanes, self.head_nums, feedforward_ratio=self.feedforward_ratio,
                                    dropout_prob=dropout_prob, act_type=self.act_type)
            for _ in range(self.encoder_layer_nums)])
        self.decoder_blocks = nn.ModuleList([
            TransformerDecoderLayer(self.inplanes, self.head_nums, feedforward_ratio=self.feedforward_ratio,
                                    dropout_prob=dropout_prob, act_type=self.act_type)
            for _ in range(self.decoder_layer_nums)])
        self.decoder_norm = nn.LayerNorm(self.inplanes)
        for m in self.parameters():
            if m.dim() > 1:
                nn.init.xavier_uniform_(m)

    def forward(self, src, mask, query_embed, pos_embed):
        """src [B, C, H, W] (NHWC memory), mask fp32 [B, H, W], query_embed [Q, C], pos_embed [B, C, H, W]
        -> hs [layers, B, Q, C], memory [B, C, H, W]."""
        b, c, h, w = src.shape
        dt = src.dtype
        src = src.flatten(2).permute(0, 2, 1)                          # [B, HW, C]: a view of the NHWC buffer
        pos_embed = pos_embed.flatten(2).permute(0, 2, 1).to(dt)
        query_pos = query_embed.to(dt).unsqueeze(0).expand(b, -1, -1)
        key_bias = mask.flatten(1).float().contiguous()                # added to the logits: +1.0 on padding
        tgt = torch.zeros_like(query_pos)
        memory = src
        for layer in self.encoder_blocks:
            memory = layer(memory, src_key_padding_mask=key_bias, pos=pos_embed)
        intermediate = []
        memory_pos = memory + pos_embed                                # the cross-attention key input of all six layers (r06: once, not six times)
        for layer in self.decoder_blocks:
            tgt = layer(tgt, memory, memory_key_padding_mask=key_bias, pos=pos_embed, query_pos=query_pos, memory_pos=memory_pos)
            intermediate.append(_ln(self.decoder_norm, tgt))
        hs = torch.stack(intermediate)                                 # [layers, B, Q, C]
        memory = memory.permute(0, 2, 1).reshape(b, c, h, w)
        return hs, memory


class _ProjConv(nn.Conv2d):
    """nn.Conv2d parameter layout, implicit-GEMM execution (1x1 projection with bias)."""

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0])


class DETR(nn.Module):

    def __init__(self, backbone_type, backbone_pretrained_path='', hidden_inplanes=256, query_nums=100, num_classes=80,
                 use_gradient_checkpoint=False):
        super(DETR, self).__init__()
        self.hidden_inplanes = hidden_inplanes
        self.query_nums = query_nums
        self.num_classes = num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.backbone = backbones.__dict__[backbone_type](**{
            'pretrained_path': backbone_pretrained_path,
            'use_gradient_checkpoint': use_gradient_checkpoint,
        })
        self.position_embedding = PositionEmbeddingBlock(inplanes=self.hidden_inplanes // 2, temperature=10000, eps=1e-6)
        self.proj_conv = _ProjConv(self.backbone.out_channels[-1], self.hidden_inplanes, kernel_size=1, stride=1,
                                   padding=0, bias=True)
        self.transformer = DETRTransformer(inplanes=self.hidden_inplanes, head_nums=8, feedforward_ratio=4,
                                           encoder_layer_nums=6, decoder_layer_nums=6, dropout_prob=0.1,
                                           act_type='relu')
        self.query_embed = nn.Embedding(self.query_nums, self.hidden_inplanes)
        self.head = DETRClsRegHead(self.hidden_inplanes, self.num_classes + 1, num_layers=3)

    def forward(self, inputs, masks):
        assert masks is not None
        features = self.backbone(inputs)[-1]
        masks = F.interpolate(masks.float().unsqueeze(1), size=[features.shape[2], features.shape[3]]).to(
            torch.bool).squeeze(1)
        positions = self.position_embedding(masks)
        features = self.proj_conv(features)
        if self.use_gradient_checkpoint:
            features, memory = checkpoint(self.transformer, features, masks.float(), self.query_embed.weight, positions,
                                          use_reentrant=False)
            cls_outputs, reg_outputs = checkpoint(self.head, features, use_reentrant=False)
        else:
            features, memory = self.transformer(features, masks.float(), self.query_embed.weight, positions)
            cls_outputs, reg_outputs = self.head(features)
        del features
        # cls_outputs [6, B, query_nums, num_classes + 1], reg_outputs [6, B, query_nums, 4]
        return [cls_outputs, reg_outputs]


def _detr(backbone_type, backbone_pretrained_path, **kwargs):
    return DETR(backbone_type, backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet18_detr(backbone_pretrained_path='', **kwargs):
    return _detr('detr_resnet18backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet34_detr(backbone_pretrained_path='', **kwargs):
    return _detr('detr_resnet34backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet50_detr(backbone_pretrained_path='', **kwargs):
    return _detr('detr_resnet50backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet101_detr(backbone_pretrained_path='', **kwargs):
    return _detr('detr_resnet101backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)


def resnet152_detr(backbone_pretrained_path='', **kwargs):
    return _detr('detr_resnet152backbone', backbone_pretrained_path=backbone_pretrained_path, **kwargs)
