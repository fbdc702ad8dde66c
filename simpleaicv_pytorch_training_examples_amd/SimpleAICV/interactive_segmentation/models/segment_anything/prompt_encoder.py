"""SAM prompt encoder -- drop-in for the reference module, built on the engine's kernels.

Interface contract (reference SimpleAICV/interactive_segmentation/models/segment_anything/prompt_encoder.py):
PositionEmbeddingRandom (:7), LayerNorm2d (:51), PromptEncoder (:69); same constructor arguments, parameter /
buffer names (`pe_layer.positional_encoding_gaussian_matrix`, `point_embeddings.N.weight`,
`not_a_point_embed.weight`, `no_mask_embed.weight`, `mask_downscaling.{0,1,3,4,6}.*`) and draw order.

Sparse path: ONE kernel (`saicv_sam_prompt_tokens`, csrc/input.hip) turns the clicks / box corners of a batch into their
tokens -- random-Fourier encoding of the pixel centre plus the learned row its kind selects, "not a point" rows replacing the
encoding -- and records each token's kind; its backward (`saicv_sam_prompt_tokens_bwd`) scatters the token gradients into
the five learned rows, the only trainable inputs of the path.  The dense position encoding of the 64 x 64 grid is the same
formula on the grid centres (`saicv_sam_grid_pe`).  Dense path: see `embed_masks`.
"""
import torch
import torch.nn as nn

from ..... import _lib, ops_tfm
from ....._lib import check, lib, ptr, require_gpu, stream


class PromptTokensFn(torch.autograd.Function):
    """tokens[B, T, C] = encoding(points | padding click | box corners) + table[kind]; gradient only to the table rows."""

    @staticmethod
    def forward(ctx, table, gauss, points, boxes, pad, image_size):
        require_gpu(table, gauss)
        src = points if points is not None else boxes
        b = src.shape[0]
        npts = points.shape[1] if points is not None else 0
        f = gauss.shape[1]
        t = (npts + (1 if (points is not None and pad) else 0)) + (2 if boxes is not None else 0)
        pts = points.detach().float().contiguous() if points is not None else None
        bxs = boxes.detach().float().reshape(b, 4).contiguous() if boxes is not None else None
        tokens = torch.empty((b, t, 2 * f), dtype=torch.float32, device=table.device)
        kinds = torch.empty((b, t), dtype=torch.int32, device=table.device)
        check(lib().saicv_sam_prompt_tokens(ptr(pts), npts, int(bool(pad and points is not None)), ptr(bxs), ptr(gauss), f,
                                            ptr(table), float(image_size), ptr(tokens), ptr(kinds), b, stream()), 'sam_prompt_tokens')
        ctx.save_for_backward(kinds)
        ctx.rows = table.shape[0]
        return tokens

    @staticmethod
    def backward(ctx, dtokens):
        kinds, = ctx.saved_tensors
        d = dtokens.float().contiguous()
        dtable = torch.zeros((ctx.rows, d.shape[-1]), dtype=torch.float32, device=d.device)
        check(lib().saicv_sam_prompt_tokens_bwd(ptr(d), ptr(kinds), ptr(dtable), kinds.numel(), d.shape[-1], stream()),
              'sam_prompt_tokens_bwd')
        return dtable, None, None, None, None, None


class PositionEmbeddingRandom(nn.Module):
    """Random-Fourier features of a 2-D position: [sin, cos](2 pi (2 p - 1) G), G ~ N(0, 1)^{2 x F} drawn at construction."""

    def __init__(self, num_pos_feats=64):
        super(PositionEmbeddingRandom, self).__init__()
        self.register_buffer("positional_encoding_gaussian_matrix", torch.randn((2, num_pos_feats)))

    def forward(self, size):
        """Encoding of the size x size grid centres: [2F, size, size] (fp32)."""
        g = self.positional_encoding_gaussian_matrix.float().contiguous()
        require_gpu(g)
        out = torch.empty((2 * g.shape[1], size, size), dtype=torch.float32, device=g.device)
        check(lib().saicv_sam_grid_pe(ptr(g), g.shape[1], size, ptr(out), stream()), 'sam_grid_pe')
        return out


class LayerNorm2d(nn.Module):

    def __init__(self, inplanes, eps=1e-6):
        super(LayerNorm2d, self).__init__()
        self.weight = nn.Parameter(torch.ones(inplanes))
        self.bias = nn.Parameter(torch.zeros(inplanes))
        self.eps = eps

    def forward(self, x):
        # channel-axis LayerNorm of an NCHW tensor on the HIP kernel (rows = pixels)
        y = ops_tfm.layer_norm(x.permute(0, 2, 3, 1), self.weight, self.bias, self.eps)
        return y.permute(0, 3, 1, 2)


class PromptEncoder(nn.Module):

    def __init__(self, image_size=1024, patch_size=16, embedding_planes=256, mask_inter_planes=16):
        super(PromptEncoder, self).__init__()
        self.image_size = image_size
        self.embedding_planes = embedding_planes
        self.image_embedding_size = image_size // patch_size
        self.pe_layer = PositionEmbeddingRandom(embedding_planes // 2)
        self.num_point_embeddings = 4            # negative / positive click, two box corners
        self.point_embeddings = nn.ModuleList(
            [nn.Embedding(1, embedding_planes) for _ in range(self.num_point_embeddings)])
        self.not_a_point_embed = nn.Embedding(1, embedding_planes)
        self.no_mask_embed = nn.Embedding(1, embedding_planes)
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, mask_inter_planes // 4, kernel_size=2, stride=2, padding=0),
            LayerNorm2d(mask_inter_planes // 4), nn.GELU(),
            nn.Conv2d(mask_inter_planes // 4, mask_inter_planes, kernel_size=2, stride=2, padding=0),
            LayerNorm2d(mask_inter_planes), nn.GELU(),
            nn.Conv2d(mask_inter_planes, embedding_planes, kernel_size=1, stride=1, padding=0))

    def _token_table(self):
        """[5, C]: the learned rows in the kernel's kind order (clicks 0 / 1, box corners, not-a-point)."""
        return torch.cat([e.weight for e in self.point_embeddings] + [self.not_a_point_embed.weight], dim=0).float()

    def sparse_tokens(self, points, boxes):
        """[B, T, C] tokens of the clicks (+ the padding click when no box comes with them) followed by the box corners."""
        gauss = self.pe_layer.positional_encoding_gaussian_matrix.float().contiguous()
        return PromptTokensFn.apply(self._token_table(), gauss, points, boxes, boxes is None, self.image_size)

    def forward(self, points, boxes, masks):
        given = [t for t in (points, boxes, masks) if t is not None]
        batch_size = given[0].shape[0] if given else 1
        if points is None and boxes is None:
            sparse = torch.empty((batch_size, 0, self.embedding_planes), device=self.no_mask_embed.weight.device)
        else:
            sparse = self.sparse_tokens(points, boxes)
        if masks is None:
            g = self.image_embedding_size
            dense = self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(batch_size, -1, g, g)
        else:
            dense = self.embed_masks(masks)
        return sparse, dense

    def get_dense_pe_layer(self):
        return self.pe_layer(self.image_embedding_size).unsqueeze(0)

    def embed_masks(self, masks):
        """mask_downscaling (reference prompt_encoder.py:93-109): two 2 x 2 stride-2 convolutions, each followed by
        LayerNorm2d + GELU, then a 1 x 1 convolution.  A 2 x 2 stride-2 convolution is a GEMM over non-overlapping
        patches: space-to-depth on NHWC, then the implicit-GEMM linear kernel (K = 4 Cin), LayerNorm over the channel
        axis and GELU on the HIP kernels -- in fp32, as the first layers see a 1-channel mask.  (Left to ATen these
        two convolutions took MIOpen's naive weight-gradient kernels: 27 ms per call, half of the SAM step's GPU
        time -- profiles/r02_samfull_rocprofv3_kernel_stats.csv.)"""
        m = self.mask_downscaling
        x = masks.float().permute(0, 2, 3, 1).contiguous()                     # [B, 256, 256, 1]
        for conv, norm in ((m[0], m[1]), (m[3], m[4])):
            b, h, w, ci = x.shape
            co = conv.weight.shape[0]
            patches = x.view(b, h // 2, 2, w // 2, 2, ci).permute(0, 1, 3, 2, 4, 5).reshape(b, h // 2, w // 2, 4 * ci)
            wm = conv.weight.permute(0, 2, 3, 1).reshape(co, 4 * ci)            # columns (di, dj, ci)
            x = ops_tfm.linear_nd(patches.contiguous(), wm, conv.bias)
            x = ops_tfm.gelu(ops_tfm.layer_norm(x, norm.weight, norm.bias, norm.eps))
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else torch.float32
        y = ops_tfm.linear_nd(x.to(dt), m[6].weight.flatten(1), m[6].bias)      # [B, 64, 64, 256]
        return y.permute(0, 3, 1, 2)
