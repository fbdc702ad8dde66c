"""Streaming glue kernels of the convolutional backbones (csrc/elemwise.hip) through the C-ABI against the
torch fp32 formulation of what the reference calls between its convolutions: nn.ReLU / nn.LeakyReLU(0.1) / nn.SiLU
(classification/backbones/darknet.py:16-33), `u * attn` (van.py:91), `x + layer_scale * f` (van.py:181-185), plain residual joins
(convformer.py:157-163), BatchNorm2d on a block input (van.py:176; training statistics, running-statistics update, eval mode) and
the per-sample stochastic-depth factor (van.py:118-150).  Operands are rounded to the compute dtype first; tolerances: fp32 1e-5
(elementwise) / 1e-4 (reductions), bf16 1e-2 of the tensor's scale (one rounding of the stored result), 2e-2 for reductions."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]
IDS = ['fp32', 'bf16']


def _rnd(dt):
    return (lambda t: t.to(torch.bfloat16).float()) if dt == torch.bfloat16 else (lambda t: t)


def _dev(t, dt):
    return t.to(dt).cuda().contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.to(dt).cuda()


@pytest.mark.parametrize('dt', DTYPES, ids=IDS)
@pytest.mark.parametrize('kind', ['relu', 'leakyrelu', 'silu'])
@pytest.mark.parametrize('shape', [(2, 16, 9, 7), (3, 40, 5, 5), (64, 24)])
def test_activations(shape, kind, dt):
    from simpleaicv_pytorch_training_examples_amd import ops
    g = torch.Generator().manual_seed(len(shape) * 7 + len(kind))
    r = _rnd(dt)
    x = r(torch.randn(shape, generator=g) * 2)
    x.view(-1)[::11] = 0.                                  # exact zeros: the ReLU family's gradient there is the negative side's
    dy = r(torch.randn(shape, generator=g))
    xr = x.clone().requires_grad_(True)
    ref = {'relu': F.relu, 'leakyrelu': lambda t: F.leaky_relu(t, 0.1), 'silu': F.silu}[kind](xr)
    ref.backward(dy)
    xd = _dev(x, dt).requires_grad_(True)
    y = ops.act(xd, kind, 0.1 if kind == 'leakyrelu' else 0.)
    y.backward(_dev(dy, dt))
    torch.cuda.synchronize()
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert y.dtype == dt and rel_err(y.float().cpu(), ref.detach()) < tol
    assert rel_err(xd.grad.float().cpu(), xr.grad) < tol


@pytest.mark.parametrize('dt', DTYPES, ids=IDS)
def test_gate_product(dt):
    from simpleaicv_pytorch_training_examples_amd import ops
    g = torch.Generator().manual_seed(3)
    r = _rnd(dt)
    a, b, dy = (r(torch.randn(2, 32, 6, 10, generator=g)) for _ in range(3))
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (ar * br).backward(dy)
    ad, bd = _dev(a, dt).requires_grad_(True), _dev(b, dt).requires_grad_(True)
    out = ops.mul(ad, bd)
    out.backward(_dev(dy, dt))
    torch.cuda.synchronize()
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert rel_err(out.float().cpu(), a * b) < tol
    assert rel_err(ad.grad.float().cpu(), ar.grad) < tol and rel_err(bd.grad.float().cpu(), br.grad) < tol


@pytest.mark.parametrize('dt', DTYPES, ids=IDS)
@pytest.mark.parametrize('scaled', [True, False], ids=['layer_scale', 'plain'])
@pytest.mark.parametrize('shape', [(2, 32, 14, 14), (3, 160, 7, 5), (1, 8, 3, 3), (2, 520, 9, 9)])
def test_scaled_residual_join(shape, scaled, dt):
    """x + s[c] * y and its three gradients (dx = dout, dy = s * dout, ds = sum dout * y), C from one chunk to more than one
    column group of the reduction (520 channels = 65 bf16 chunks)"""
    from simpleaicv_pytorch_training_examples_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    r = _rnd(dt)
    x, y, dout = (r(torch.randn(shape, generator=g)) for _ in range(3))
    s = torch.randn(1, shape[1], 1, 1, generator=g)
    xr, yr, sr = x.clone().requires_grad_(True), y.clone().requires_grad_(True), s.clone().requires_grad_(True)
    ref = xr + (sr * yr if scaled else yr)
    ref.backward(dout)
    xd, yd = _dev(x, dt).requires_grad_(True), _dev(y, dt).requires_grad_(True)
    sd = s.cuda().requires_grad_(True)
    out = ops.scale_add(xd, yd, sd if scaled else None)
    out.backward(_dev(dout, dt))
    torch.cuda.synchronize()
    f32 = dt == torch.float32
    assert rel_err(out.float().cpu(), ref.detach()) < (1e-5 if f32 else 1e-2)
    assert rel_err(xd.grad.float().cpu(), xr.grad) < (1e-6 if f32 else 1e-2)
    assert rel_err(yd.grad.float().cpu(), yr.grad) < (1e-5 if f32 else 1e-2)
    if scaled:
        assert sd.grad.shape == s.shape and rel_err(sd.grad.cpu(), sr.grad) < (1e-4 if f32 else 2e-2)


@pytest.mark.parametrize('dt', DTYPES, ids=IDS)
@pytest.mark.parametrize('shape', [(4, 32, 12, 12), (2, 160, 7, 9), (8, 8, 5, 5), (2, 520, 6, 6)])
def test_batchnorm_on_a_block_input(shape, dt):
    """training mode: output, input / weight / bias gradients, running statistics and num_batches_tracked after two steps; then
    eval mode on the updated running statistics"""
    from simpleaicv_pytorch_training_examples_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + 1)
    r = _rnd(dt)
    c = shape[1]
    ref_bn, bn = nn.BatchNorm2d(c), nn.BatchNorm2d(c)
    with torch.no_grad():
        ref_bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        ref_bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
    bn.load_state_dict(ref_bn.state_dict())
    bn.cuda()
    f32 = dt == torch.float32
    for step in range(2):
        x = r(torch.randn(shape, generator=g) * 1.5 + 0.3)
        dz = r(torch.randn(shape, generator=g))
        xr = x.clone().requires_grad_(True)
        ref = ref_bn(xr)
        ref_bn.zero_grad()
        ref.backward(dz)
        xd = _dev(x, dt).requires_grad_(True)
        bn.zero_grad()
        z = ops.batch_norm2d(xd, bn)
        z.backward(_dev(dz, dt))
        torch.cuda.synchronize()
        assert z.dtype == dt and rel_err(z.float().cpu(), ref.detach()) < (1e-4 if f32 else 1e-2)
        assert rel_err(xd.grad.float().cpu(), xr.grad) < (2e-4 if f32 else 2e-2)
        assert rel_err(bn.weight.grad.cpu(), ref_bn.weight.grad) < (1e-4 if f32 else 2e-2)
        assert rel_err(bn.bias.grad.cpu(), ref_bn.bias.grad) < (1e-4 if f32 else 2e-2)
    assert int(bn.num_batches_tracked) == 2
    assert rel_err(bn.running_mean.cpu(), ref_bn.running_mean) < 1e-4 and rel_err(bn.running_var.cpu(), ref_bn.running_var) < 1e-4
    ref_bn.eval()
    bn.eval()
    x = r(torch.randn(shape, generator=g))
    xr = x.clone().requires_grad_(True)
    ref = ref_bn(xr)
    dz = r(torch.randn(shape, generator=g))
    ref_bn.zero_grad()
    ref.backward(dz)
    xd = _dev(x, dt).requires_grad_(True)
    bn.zero_grad()
    z = ops.batch_norm2d(xd, bn)
    z.backward(_dev(dz, dt))
    assert int(bn.num_batches_tracked) == 2
    assert rel_err(z.float().cpu(), ref.detach()) < (1e-4 if f32 else 1e-2)
    assert rel_err(xd.grad.float().cpu(), xr.grad) < (1e-4 if f32 else 1e-2)
    # frozen statistics still train the affine parameters (torch: dgamma = sum dz * xhat, dbeta = sum dz in eval mode too)
    assert bn.weight.grad is not None and bn.bias.grad is not None
    assert rel_err(bn.weight.grad.cpu(), ref_bn.weight.grad) < (2e-4 if f32 else 2e-2)
    assert rel_err(bn.bias.grad.cpu(), ref_bn.bias.grad) < (1e-4 if f32 else 2e-2)
    # ... and a frozen layer (requires_grad off) keeps the cheap path: input gradient only
    for q in bn.parameters():
        q.requires_grad_(False)
    xd2 = _dev(x, dt).requires_grad_(True)
    ops.batch_norm2d(xd2, bn).backward(_dev(dz, dt))
    assert rel_err(xd2.grad.float().cpu(), xr.grad) < (1e-4 if f32 else 1e-2)


@pytest.mark.parametrize('dt', DTYPES, ids=IDS)
def test_per_sample_factor(dt):
    from simpleaicv_pytorch_training_examples_amd import ops
    g = torch.Generator().manual_seed(5)
    r = _rnd(dt)
    x, dy = r(torch.randn(4, 24, 5, 7, generator=g)), r(torch.randn(4, 24, 5, 7, generator=g))
    w = torch.tensor([0., 1.25, 1.25, 0.])
    xd = _dev(x, dt).requires_grad_(True)
    out = ops.sample_scale(xd, w.cuda())
    out.backward(_dev(dy, dt))
    tol = 1e-6 if dt == torch.float32 else 1e-2
    assert rel_err(out.float().cpu(), x * w.view(4, 1, 1, 1)) < tol
    assert rel_err(xd.grad.float().cpu(), dy * w.view(4, 1, 1, 1)) < tol


def test_statistics_pass_at_a_stage_sized_activation_is_additive():
    """Size-independent property at a BASELINE-sized activation (batch 128 of VAN-B0's first stage, 32 x 56 x 56, bf16): the
    statistics of the batch equal the sum of the statistics of its halves, and match an fp64 sum of the same bf16 values."""
    import ctypes
    from simpleaicv_pytorch_training_examples_amd import _lib
    L = _lib.lib()
    x = (torch.randn(128, 56, 56, 32, device='cuda') * 2 + 0.5).to(torch.bfloat16)
    m = 128 * 56 * 56

    def stats(t, rows):
        s = torch.zeros(2, 32, device='cuda')
        _lib.check(L.saicv_bn_stats(_lib.BF16, t.data_ptr(), rows, 32, s[0].data_ptr(), s[1].data_ptr(), None), 'bn_stats')
        return s

    full, lo, hi = stats(x, m), stats(x[:64], m // 2), stats(x[64:], m // 2)
    torch.cuda.synchronize()
    assert rel_err(full, lo + hi) < 1e-5
    xd = x.double().view(-1, 32)
    assert rel_err(full[0], xd.sum(0)) < 1e-4 and rel_err(full[1], (xd * xd).sum(0)) < 1e-4


@pytest.mark.parametrize('dt', DTYPES, ids=IDS)
@pytest.mark.parametrize('relu', [False, True], ids=['plain', 'relu'])
@pytest.mark.parametrize('shape,groups', [((2, 256, 12, 10), 32), ((3, 64, 5, 7), 32), ((1, 32, 3, 3), 4), ((2, 96, 9, 9), 3)])
def test_group_norm_on_nhwc(shape, groups, relu, dt):
    """csrc/groupnorm.hip against F.group_norm (+ F.relu) in fp32 on the same (rounded) operands: output, input gradient,
    dgamma, dbeta; groups of one chunk (FCOS: 256 channels / 32 groups), of two bf16 chunks... and of 32 channels"""
    from simpleaicv_pytorch_training_examples_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + groups)
    r = _rnd(dt)
    c = shape[1]
    gn = nn.GroupNorm(groups, c)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(c, generator=g) * 0.3)
    x = r(torch.randn(shape, generator=g) * 1.7 + 0.4)
    dy = r(torch.randn(shape, generator=g))
    xr = x.clone().requires_grad_(True)
    ref = F.group_norm(xr, groups, gn.weight, gn.bias, gn.eps)
    if relu:
        ref = F.relu(ref)
    ref.backward(dy)
    wr, br = gn.weight.grad.clone(), gn.bias.grad.clone()
    gn.zero_grad()
    gn.cuda()
    xd = _dev(x, dt).requires_grad_(True)
    y = ops.group_norm(xd, gn, relu=relu)
    y.backward(_dev(dy, dt))
    torch.cuda.synchronize()
    f32 = dt == torch.float32
    assert y.dtype == dt and rel_err(y.float().cpu(), ref.detach()) < (1e-4 if f32 else 1e-2)
    assert rel_err(xd.grad.float().cpu(), xr.grad) < (2e-4 if f32 else 2e-2)
    assert rel_err(gn.weight.grad.cpu(), wr) < (1e-4 if f32 else 2e-2)
    assert rel_err(gn.bias.grad.cpu(), br) < (1e-4 if f32 else 2e-2)
