"""SAM prompt encoder -- drop-in for the reference module.

Interface contract (reference SimpleAICV/interactive_segmentation/models/segment_anything/prompt_encoder.py):
PositionEmbeddingRandom (:7), LayerNorm2d (:51), PromptEncoder (:69); same constructor arguments, parameter /
buffer names (`pe_layer.positional_encoding_gaussian_matrix`, `point_embeddings.N.weight`,
`not_a_point_embed.weight`, `no_mask_embed.weight`, `mask_downscaling.{0,1,3,4,6}.*`) and draw order.

The sparse path (random-Fourier encoding of <= a dozen points / box corners per sample) is host-scale
tensor glue and stays in plain tensor ops.  In the dense path the two 2x2 stride-2 convs run on 1 and 4
channels -- below the 16-byte chunk the implicit-GEMM kernels stream -- so they, their LayerNorm2d and GELU
stay as tensor ops on [B, <=16, <=128, <=128]; the 16 -> 256 projection onto the 64x64 grid is the HIP linear.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..... import ops_tfm


class PositionEmbeddingRandom(nn.Module):

    def __init__(self, num_pos_feats=64):
        super(PositionEmbeddingRandom, self).__init__()
        self.register_buffer("positional_encoding_gaussian_matrix", torch.randn((2, num_pos_feats)))

    def forward(self, size):
        """Positional encoding of a size x size grid: C x H x W."""
        h, w = size, size
        device = self.positional_encoding_gaussian_matrix.device
        grid = torch.ones((h, w), device=device, dtype=torch.float32)
        y_embed = (grid.cumsum(dim=0) - 0.5) / h
        x_embed = (grid.cumsum(dim=1) - 0.5) / w
        pe = self.pe_encoding(torch.stack([x_embed, y_embed], dim=-1))
        return pe.permute(2, 0, 1)

    def forward_with_coords(self, coords_input, image_size):
        """Positionally encode points that are not normalized to [0,1]."""
        coords = coords_input.clone()
        coords[:, :, 0] = coords[:, :, 0] / image_size
        coords[:, :, 1] = coords[:, :, 1] / image_size
        return self.pe_encoding(coords.to(torch.float))

    def pe_encoding(self, coords):
        coords = 2 * coords - 1
        with torch.autocast(coords.device.type, enabled=False):      # tiny K=2 product, keep fp32 phases
            coords = coords.float() @ self.positional_encoding_gaussian_matrix.float()
        coords = 2 * np.pi * coords
        return torch.cat([torch.sin(coords), torch.cos(coords)], dim=-1)


class LayerNorm2d(nn.Module):

    def __init__(self, inplanes, eps=1e-6):
        super(LayerNorm2d, self).__init__()
        self.weight = nn.Parameter(torch.ones(inplanes))
        self.bias = nn.Parameter(torch.zeros(inplanes))
        self.eps = eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class PromptEncoder(nn.Module):

    def __init__(self, image_size=1024, patch_size=16, embedding_planes=256, mask_inter_planes=16):
        super(PromptEncoder, self).__init__()
        self.image_size = image_size
        self.embedding_planes = embedding_planes
        self.image_embedding_size = image_size // patch_size
        self.pe_layer = PositionEmbeddingRandom(embedding_planes // 2)
        # pos/neg point + 2 box corners
        self.num_point_embeddings = 4
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

    def forward(self, points, boxes, masks):
        if points is not None:
            batch_size = points.shape[0]
        elif boxes is not None:
            batch_size = boxes.shape[0]
        elif masks is not None:
            batch_size = masks.shape[0]
        else:
            batch_size = 1
        device = self.point_embeddings[0].weight.device
        sparse_embeddings = torch.empty((batch_size, 0, self.embedding_planes), device=device)
        if points is not None:
            coords, labels = points[:, :, 0:2], points[:, :, 2]
            point_embeddings = self.embed_points(coords, labels, pad=(boxes is None))
            sparse_embeddings = torch.cat([sparse_embeddings, point_embeddings], dim=1)
        if boxes is not None:
            sparse_embeddings = torch.cat([sparse_embeddings, self.embed_boxes(boxes)], dim=1)
        if masks is not None:
            dense_embeddings = self.embed_masks(masks)
        else:
            dense_embeddings = self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(
                batch_size, -1, self.image_embedding_size, self.image_embedding_size)
        return sparse_embeddings, dense_embeddings

    def get_dense_pe_layer(self):
        return self.pe_layer(self.image_embedding_size).unsqueeze(0)

    def embed_points(self, points, labels, pad):
        points = points + 0.5                                   # shift to the pixel centre
        if pad:
            points = torch.cat([points, torch.zeros((points.shape[0], 1, 2), device=points.device)], dim=1)
            labels = torch.cat([labels, -torch.ones((labels.shape[0], 1), device=labels.device)], dim=1)
        point_embedding = self.pe_layer.forward_with_coords(points, self.image_size)
        point_embedding[labels == -1] = 0.0
        point_embedding[labels == -1] += self.not_a_point_embed.weight
        point_embedding[labels == 0] += self.point_embeddings[0].weight
        point_embedding[labels == 1] += self.point_embeddings[1].weight
        return point_embedding

    def embed_boxes(self, boxes):
        boxes = boxes + 0.5
        coords = boxes.reshape(-1, 2, 2)
        corner_embedding = self.pe_layer.forward_with_coords(coords, self.image_size)
        corner_embedding[:, 0, :] += self.point_embeddings[2].weight
        corner_embedding[:, 1, :] += self.point_embeddings[3].weight
        return corner_embedding

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
