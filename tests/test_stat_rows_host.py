"""Host-side contract of the BN partial-statistics buffer (no GPU): `saicv_conv2d_stat_rows` is what the Python
wrapper sizes the [2, rows, K] buffer with BEFORE the convolution launch picks its tile geometry, so it must be a
pure function of the descriptor and agree with one row per (row of workgroups, row of wavefronts) of one of the
geometries: since r03 every WAVEFRONT writes (or atomically adds) the column sums of its own rows of the tile."""
import ctypes

import pytest
import torch

from simpleaicv_pytorch_training_examples_amd import ops
from simpleaicv_pytorch_training_examples_amd._lib import lib

# (Cin, Cout, k, stride, H) of ResNet-50 at 224 x 224
SHAPES = [(8, 64, 7, 2, 224), (64, 64, 1, 1, 56), (64, 256, 1, 1, 56), (256, 128, 1, 1, 56), (128, 128, 3, 2, 56),
          (512, 1024, 1, 2, 28), (256, 256, 3, 1, 14), (512, 2048, 1, 1, 7), (512, 512, 3, 1, 7)]


@pytest.mark.parametrize('batch', [2, 256])
@pytest.mark.parametrize('ci,co,k,s,h', SHAPES)
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_stat_rows_is_one_row_per_wavefront_row(ci, co, k, s, h, batch, dt):
    d = ops._desc(batch, h, h, ci, co, k, k, s, k // 2, dt)
    L = lib()
    rows = L.saicv_conv2d_stat_rows(ctypes.byref(d))
    m = batch * d.OH * d.OW
    # one row per tile row (one tile per workgroup), or per (tile row, wavefront row) in a persistent launch:
    # 256 x 256 (2 wavefront rows), 256 x 128 on four wavefronts (2), 128 x 128 (2), 128 x 64 (2)
    # r06: pointwise stride-1 products of the shapes of csrc/pwstream.hip over >= 65 536 rows run as ONE resident round of the streaming kernel
    # (csrc/pwstream.hip): one row per workgroup, 2 workgroups per CU
    streamed = dt == torch.bfloat16 and k == 1 and s == 1 and m >= 65536 and (ci, co) in {(64, 64), (64, 256)}
    streamed3 = dt == torch.bfloat16 and k == 3 and s == 1 and m >= 65536 and (ci, co) == (64, 64)      # the same stream over nine taps: one workgroup per CU
    if streamed or streamed3:
        assert rows == (512 if streamed else 256), (rows, m)
    else:
        assert rows in {-(-m // 256), -(-m // 128), -(-m // 256) * 2, -(-m // 128) * 2}, (rows, m)
    assert rows == L.saicv_conv2d_stat_rows(ctypes.byref(d))          # pure function of the descriptor
