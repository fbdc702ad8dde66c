"""Anchor boxes (RetinaNet) and point positions (FCOS) of one image's pyramid levels -- the host-side tables of reference
SimpleAICV/detection/models/anchor.py (RetinaAnchors :5-86, FCOSPositions :89-131), same constructor arguments and output
contract: a list over the levels of float32 arrays [h, w, anchors, 4] (x_min, y_min, x_max, y_max) / [h, w, 2] (x, y centres),
`fpn_feature_sizes[level] = [w, h]`.  Built with broadcasting instead of nested Python lists; values are bit-identical to the
reference's (tests/test_host_logic.py pins them against reference-generated fixtures)."""
import math

import numpy as np


def _centres(size, stride):
    """pixel-centre coordinates of `size` cells at `stride`; float64 like the reference's arange arithmetic, rounded at the end"""
    return (np.arange(0, size) + 0.5) * stride


class RetinaAnchors:

    def __init__(self, areas=[[32, 32], [64, 64], [128, 128], [256, 256], [512, 512]], ratios=[0.5, 1, 2],
                 scales=[2**0, 2**(1.0 / 3.0), 2**(2.0 / 3.0)], strides=[8, 16, 32, 64, 128]):
        self.areas = np.array(areas, dtype=np.float32)
        self.ratios = np.array(ratios, dtype=np.float32)
        self.scales = np.array(scales, dtype=np.float32)
        self.strides = np.array(strides, dtype=np.float32)

    def __call__(self, fpn_feature_sizes):
        return [self.generate_anchors_on_feature_map(self.generate_base_anchors(area, self.scales, self.ratios),
                                                     fpn_feature_sizes[level], self.strides[level])
                for level, area in enumerate(self.areas)]

    def generate_base_anchors(self, area, scales, ratios):
        """[len(ratios) * len(scales), 4] boxes centred on the origin; ratio-major like the reference (:43-45)"""
        aspects = np.array([[s * math.sqrt(r), s * math.sqrt(1 / r)] for r in ratios for s in scales], dtype=np.float32)
        wh = area * aspects                                       # float32 [A, 2]
        half = wh / 2
        return np.concatenate([0 - half, half], axis=1).astype(np.float32)

    def generate_anchors_on_feature_map(self, base_anchors, feature_map_size, stride):
        w, h = feature_map_size
        cx = _centres(w, stride).astype(np.float32)
        cy = _centres(h, stride).astype(np.float32)
        shifts = np.empty((h, w, 1, 4), dtype=np.float32)
        shifts[..., 0, 0] = shifts[..., 0, 2] = cx[None, :]
        shifts[..., 0, 1] = shifts[..., 0, 3] = cy[:, None]
        return np.ascontiguousarray(shifts + base_anchors[None, None], dtype=np.float32)


class FCOSPositions:

    def __init__(self, strides=[8, 16, 32, 64, 128]):
        self.strides = np.array(strides, dtype=np.float32)

    def __call__(self, fpn_feature_sizes):
        return [self.generate_positions_on_feature_map(size, stride) for stride, size in zip(self.strides, fpn_feature_sizes)]

    def generate_positions_on_feature_map(self, feature_map_size, stride):
        w, h = feature_map_size
        out = np.empty((h, w, 2), dtype=np.float32)
        out[..., 0] = _centres(w, stride).astype(np.float32)[None, :]
        out[..., 1] = _centres(h, stride).astype(np.float32)[:, None]
        return out
