"""SAM mask decoder on the MI355X HIP kernels -- drop-in for the reference module.

Interface contract (reference SimpleAICV/interactive_segmentation/models/segment_anything/mask_decoder.py):
LayerNorm2d (:19), MLP (:36), MaskDecoder (:53); same constructor arguments and parameter names
(`transformer.*`, `iou_token.weight`, `mask_tokens.weight`, `output_upscaling.{0,1,3}.*`,
`output_hypernetworks_mlps.N.layers.M.*`, `iou_prediction_head.layers.M.*`).

Execution (NHWC throughout): a ConvTranspose2d(k=2, s=2) is a per-pixel linear map onto 4 * Cout values
followed by a pixel shuffle, so both upscaling stages are the implicit-GEMM linear on [B*H*W, Cin] rows
(+ one shuffle copy); LayerNorm2d is the channel-axis row LayerNorm; GELU the HIP elementwise kernel.
The 4 hyper-network MLPs and the IoU head act on <= 5 tokens per sample.
"""
import torch
import torch.nn as nn

from ..... import ops_tfm
from .transformer import TwoWayTransformer


class LayerNorm2d(nn.Module):

    def __init__(self, inplanes, eps=1e-6):
        super(LayerNorm2d, self).__init__()
        self.weight = nn.Parameter(torch.ones(inplanes))
        self.bias = nn.Parameter(torch.zeros(inplanes))
        self.eps = eps

    def forward(self, x):
        """x: NCHW-shaped tensor over NHWC memory (as every activation of this decoder is)."""
        t = ops_tfm.layer_norm(x.permute(0, 2, 3, 1), self.weight, self.bias, self.eps)
        return t.permute(0, 3, 1, 2)


class MLP(nn.Module):

    def __init__(self, inplanes, hidden_planes, planes, layer_nums):
        super(MLP, self).__init__()
        self.layer_nums = layer_nums
        h = [hidden_planes] * (layer_nums - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([inplanes] + h, h + [planes]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = ops_tfm.linear_nd(x, layer.weight, layer.bias)
            if i < self.layer_nums - 1:
                x = torch.relu(x)
        return x


def _conv_transpose_2x2(x_nhwc, deconv):
    """nn.ConvTranspose2d(kernel 2, stride 2) on [B, H, W, Cin] -> [B, 2H, 2W, Cout]."""
    b, h, w, ci = x_nhwc.shape
    co = deconv.weight.shape[1]
    wm = deconv.weight.permute(2, 3, 1, 0).reshape(4 * co, ci)          # rows (di, dj, co)
    bm = deconv.bias.repeat(4) if deconv.bias is not None else None
    y = ops_tfm.linear_nd(x_nhwc, wm, bm)                                # [B, H, W, 4*Cout]
    return y.view(b, h, w, 2, 2, co).permute(0, 1, 3, 2, 4, 5).reshape(b, 2 * h, 2 * w, co)


_INDEX_CACHE = {}


class MaskDecoder(nn.Module):

    def __init__(self, inplanes=256, num_multimask_outputs=3, iou_prediction_head_block_nums=3,
                 iou_prediction_head_hidden_planes=256):
        super(MaskDecoder, self).__init__()
        self.transformer = TwoWayTransformer(block_nums=2, embedding_planes=inplanes, head_nums=8, mlp_planes=2048)
        self.num_multimask_outputs = num_multimask_outputs
        self.num_mask_tokens = num_multimask_outputs + 1
        self.iou_token = nn.Embedding(1, inplanes)
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, inplanes)
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(inplanes, inplanes // 4, kernel_size=2, stride=2, padding=0),
            LayerNorm2d(inplanes // 4), nn.GELU(),
            nn.ConvTranspose2d(inplanes // 4, inplanes // 8, kernel_size=2, stride=2, padding=0), nn.GELU())
        self.output_hypernetworks_mlps = nn.ModuleList(
            [MLP(inplanes, inplanes, inplanes // 8, 3) for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = MLP(inplanes=inplanes, hidden_planes=iou_prediction_head_hidden_planes,
                                       planes=self.num_mask_tokens, layer_nums=iou_prediction_head_block_nums)

    def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings,
                mask_out_idxs=[0, 1, 2, 3]):
        output_tokens = torch.cat([self.iou_token.weight, self.mask_tokens.weight], dim=0)
        output_tokens = output_tokens.unsqueeze(0).expand(sparse_prompt_embeddings.size(0), -1, -1)
        tokens = torch.cat((output_tokens, sparse_prompt_embeddings), dim=1)

        if image_embeddings.shape[0] != tokens.shape[0]:        # one image feature for several prompts
            src = torch.repeat_interleave(image_embeddings, tokens.shape[0], dim=0)
        else:
            src = image_embeddings
        src = (src + dense_prompt_embeddings.to(src.dtype)).contiguous(memory_format=torch.channels_last)
        pos_src = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
        b, c, h, w = src.shape

        hs, src = self.transformer(src, pos_src, tokens)        # src: [B, HW, C] tokens
        iou_token_out = hs[:, 0, :]
        mask_tokens_out = hs[:, 1:(1 + self.num_mask_tokens), :]

        up = self.output_upscaling
        x = _conv_transpose_2x2(src.reshape(b, h, w, c), up[0])                 # [B, 2H, 2W, C/4]
        x = ops_tfm.gelu(ops_tfm.layer_norm(x, up[1].weight, up[1].bias, up[1].eps))
        x = ops_tfm.gelu(_conv_transpose_2x2(x, up[3]))                         # [B, 4H, 4W, C/8]
        hyper_in = torch.stack([self.output_hypernetworks_mlps[i](mask_tokens_out[:, i, :])
                                for i in range(self.num_mask_tokens)], dim=1)   # [B, T, C/8]
        b, h4, w4, c8 = x.shape
        # masks[b, t, y, x] = <hyper_in[b, t, :], upscaled[b, y, x, :]>  -- a K = 32 product, HBM-bound: one streaming
        # HIP kernel each way (csrc/samtail.hip) instead of a skinny batched GEMM plus a permute copy
        mask_preds = ops_tfm.hyper_product(x.view(b, h4 * w4, c8), hyper_in).view(b, -1, h4, w4)
        iou_preds = self.iou_prediction_head(iou_token_out)
        # (reference mask_decoder.py: mask_preds[:, mask_out_idxs]; a python list as an index is a host -> device copy per call, which a
        # captured step cannot contain: every output = no selection, a contiguous run = a slice, anything else = a cached index tensor)
        idxs = [int(i) for i in mask_out_idxs]
        if idxs == list(range(mask_preds.shape[1])):
            return mask_preds, iou_preds
        if idxs == list(range(idxs[0], idxs[0] + len(idxs))):
            return mask_preds[:, idxs[0]:idxs[0] + len(idxs)], iou_preds[:, idxs[0]:idxs[0] + len(idxs)]
        key = (tuple(idxs), mask_preds.device)
        sel = _INDEX_CACHE.get(key)
        if sel is None:
            sel = _INDEX_CACHE[key] = torch.tensor(idxs, dtype=torch.long, device=mask_preds.device)
        return mask_preds.index_select(1, sel), iou_preds.index_select(1, sel)
