"""Depthwise convolution kernels (csrc/dwconv.hip) through the C-ABI against F.conv2d(groups=C) and its
autograd in fp32 on the CPU, on the same (bf16-rounded) operands.  Geometries: the depthwise layers of reference
classification/backbones/van.py:30 (3x3 s1 p1), :68 (5x5 p2), :75 (7x7 p9 dilation 3) and convformer.py (7x7 p3), a strided 3x3 and
ragged sizes (W not a multiple of the four-pixel thread tile, C = 8).  Tolerance: fp32 1e-4 (another summation order), bf16 2e-2 of
the tensor's scale for stored activations, 1e-2 for the fp32 weight / bias gradients (sums of bf16 products)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

# (N, C, H, W, k, stride, pad, dilation)
CASES = [(2, 64, 56, 56, 3, 1, 1, 1), (2, 64, 28, 28, 5, 1, 2, 1), (2, 64, 28, 28, 7, 1, 9, 3), (2, 128, 14, 14, 7, 1, 3, 1),
         (3, 32, 17, 23, 3, 2, 1, 1), (1, 8, 9, 7, 5, 2, 2, 1), (2, 96, 12, 12, 7, 2, 3, 1), (1, 16, 5, 5, 1, 1, 0, 1)]


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('case', CASES, ids=[f'n{c[0]}c{c[1]}_{c[2]}x{c[3]}_k{c[4]}s{c[5]}p{c[6]}d{c[7]}' for c in CASES])
def test_depthwise_conv_matches_cpu_fp32(case, dt):
    from simpleaicv_pytorch_training_examples_amd import ops
    n, c, h, w, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case))
    rnd = (lambda t: t.to(torch.bfloat16).float()) if dt == torch.bfloat16 else (lambda t: t)
    x = rnd(torch.randn(n, c, h, w, generator=g))
    wt = rnd(torch.randn(c, 1, k, k, generator=g) / k)
    b = torch.randn(c, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, s, p, d, groups=c)
    dy = rnd(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)

    xd = x.to(dt).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = wt.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    y = ops.depthwise_conv2d(xd, wd, bd, s, p, d)
    assert tuple(y.shape) == tuple(y_ref.shape) and y.dtype == dt
    y.backward(dy.to(dt).cuda().contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    f32 = dt == torch.float32
    assert rel_err(y.float().cpu(), y_ref.detach()) < (1e-4 if f32 else 2e-2)
    assert rel_err(xd.grad.float().cpu(), xr.grad) < (1e-4 if f32 else 2e-2)
    assert rel_err(wd.grad.cpu(), wr.grad) < (1e-4 if f32 else 1e-2)
    assert rel_err(bd.grad.cpu(), br.grad) < (1e-4 if f32 else 1e-2)


def test_depthwise_conv_at_a_van_stage_shape_is_linear_and_additive():
    """Size-independent properties at a BASELINE-sized activation (batch 64 of VAN-B0's first stage: 32 x 56 x 56): the output
    is linear in the input, and the weight gradient over the batch is the sum of the gradients over its halves."""
    from simpleaicv_pytorch_training_examples_amd import ops
    g = torch.Generator().manual_seed(3)
    n, c, hw, k = 64, 32, 56, 5
    x = torch.randn(n, c, hw, hw, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    x2 = torch.randn(n, c, hw, hw, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, 1, k, k, generator=g) / k).cuda().requires_grad_(True)
    y1, y2, y12 = ops.depthwise_conv2d(x, w, None, 1, 2), ops.depthwise_conv2d(x2, w, None, 1, 2), ops.depthwise_conv2d(x + 2 * x2, w, None, 1, 2)
    assert rel_err(y12, y1 + 2 * y2) < 1e-5
    dy = torch.randn(y1.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    full, = torch.autograd.grad(ops.depthwise_conv2d(x, w, None, 1, 2), w, dy)
    lo, = torch.autograd.grad(ops.depthwise_conv2d(x[:32], w, None, 1, 2), w, dy[:32])
    hi, = torch.autograd.grad(ops.depthwise_conv2d(x[32:], w, None, 1, 2), w, dy[32:])
    assert rel_err(full, lo + hi) < 1e-4
