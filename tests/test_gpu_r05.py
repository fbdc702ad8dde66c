"""Round-5 kernels against their predecessors and against torch fp32: the two-rows-per-wavefront LayerNorm (default for rows of 96
chunks; SAICV_LN_HALF=0 = the one-row kernels), the convolution weight gradient with carried operand offsets (the only convolution
form of igemm_tn_dma_kernel since r05; SAICV_TN_DMA=0 = the register-staged kernel), and the r05 forward / data-gradient K loop
(assembly fragment reads, bias in the accumulators) on shapes that exercise every tile geometry and tail."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu]


@pytest.mark.parametrize('rows', [1, 2, 7, 16, 197 * 8 + 1, 50432])
@pytest.mark.parametrize('with_addend', [False, True])
def test_layernorm_two_rows_per_wavefront_matches_the_default_kernels(rows, with_addend):
    """layernorm_{fwd,bwd}_half_kernel (C = 768 bf16, the default): the same y / mean / rstd / dx / dgamma / dbeta as the one-row kernels
    (SAICV_LN_HALF=0) up to fp32 summation order, and both against torch's fp32 LayerNorm; odd row counts leave the last wavefront half empty."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    g = torch.Generator().manual_seed(rows)
    c = 768
    x = torch.randn(rows, c, generator=g).cuda().bfloat16()
    dy = torch.randn(rows, c, generator=g).cuda().bfloat16()
    add = torch.randn(rows, c, generator=g).cuda().bfloat16() if with_addend else None
    w = (torch.randn(c, generator=g) * 0.2 + 1.0).cuda()
    b = (torch.randn(c, generator=g) * 0.1).cuda()
    out = {}
    for flag in ('0', '1'):
        os.environ['SAICV_LN_HALF'] = flag
        try:
            y, mean, rstd = ops_tfm.ln_fwd(x, w, b, 1e-6)
            dx, dw, db = ops_tfm.ln_bwd(dy, x, w, b, mean, rstd, addend=add)
            torch.cuda.synchronize()
        finally:
            os.environ.pop('SAICV_LN_HALF', None)
        out[flag] = [t.float().cpu() for t in (y, mean, rstd, dx, dw, db)]
    xr = x.float().cpu().requires_grad_(True)
    wr, br = w.cpu().requires_grad_(True), b.cpu().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (c,), wr, br, 1e-6)
    yr.backward(dy.float().cpu())
    ref = [yr.detach(), None, None, xr.grad + (add.float().cpu() if with_addend else 0), wr.grad, br.grad]
    names = ('y', 'mean', 'rstd', 'dx', 'dgamma', 'dbeta')
    for name, a0, a1, r in zip(names, out['0'], out['1'], ref):
        scale = float(a0.abs().max()) + 1e-12
        tol = 1e-2 if name in ('y', 'dx') else 2e-5 * max(1.0, rows ** 0.5)      # bf16 outputs may round differently at 1 ulp
        assert float((a0 - a1).abs().max()) <= tol * scale, (name, rows, float((a0 - a1).abs().max()), scale)
        if r is not None:
            assert float((a1 - r).abs().max()) <= (2e-2 if name in ('y', 'dx') else 2e-3) * (float(r.abs().max()) + 1e-12), (name, rows)


@pytest.mark.parametrize('geom', [(8, 64, 56, 56, 64, 3, 1, 1), (8, 64, 56, 56, 128, 3, 2, 1), (4, 256, 28, 28, 512, 1, 2, 0),
                                  (2, 16, 37, 41, 24, 3, 1, 1), (3, 8, 19, 23, 16, 7, 2, 3), (16, 512, 7, 7, 512, 3, 1, 1)])
def test_wgrad_with_carried_offsets_matches_the_default_kernel(geom):
    """igemm_tn_dma_kernel, convolution form (carried offsets): the weight gradient of a bf16 convolution equals the register-staged
    kernel's (SAICV_TN_DMA=0) up to the order of the fp32 atomics, and both equal torch's fp32 convolution gradient at bf16 resolution."""
    from simpleaicv_pytorch_training_examples_amd import ops
    n, ci, h, w, co, k, stride, pad = geom
    g = torch.Generator().manual_seed(sum(geom))
    x = torch.randn(n, h, w, ci, generator=g).permute(0, 3, 1, 2).cuda().bfloat16()
    wt = (torch.randn(co, ci, k, k, generator=g) * 0.1)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, oh, ow, co, generator=g).permute(0, 3, 1, 2).cuda().bfloat16()
    grads = {}
    for flag in ('0', '1'):
        os.environ['SAICV_TN_DMA'] = flag
        try:
            wp = wt.clone().cuda().requires_grad_(True)
            ops.bump_weights_epoch()
            y = ops.conv2d(x.clone().requires_grad_(True), wp, None, stride, pad)
            y.backward(dy)
            torch.cuda.synchronize()
        finally:
            os.environ.pop('SAICV_TN_DMA', None)
        grads[flag] = wp.grad.float().cpu()
    ref = torch.nn.grad.conv2d_weight(x.float().cpu(), wt.shape, dy.float().cpu(), stride=stride, padding=pad)
    scale = float(ref.abs().max())
    assert float((grads['0'] - grads['1']).abs().max()) <= 1e-4 * scale, geom
    assert float((grads['1'] - ref).abs().max()) <= 2e-2 * scale, geom


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'fp32'])
@pytest.mark.parametrize('mkn', [(300, 96, 40), (1000, 64, 100), (513, 200, 1000), (4096, 768, 768), (257, 128, 36), (129, 32, 30),
                                 (50432, 768, 2304), (5000, 3072, 768)])
def test_linear_forward_and_input_gradient_with_bias_match_torch(mkn, dtype):
    """igemm_nt1_kernel after r05 (fragment reads as assembly statements released by counted waits, the bias as the accumulators'
    start value when N % 4 == 0, the epilogue's masked loads otherwise -- N = 30): y = x W^T + b and dx = dy W against torch fp32 on
    shapes that take every tile geometry (128 x 64 ... 256 x 256 with 128-byte K slices), with M, N and K tails."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    m, k, n = mkn
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype)
    b = torch.randn(n, generator=g)
    dy = torch.randn(m, n, generator=g).to(dtype)
    xg = x.cuda().requires_grad_(True)
    wg = w.float().cuda().requires_grad_(True)       # parameters are fp32 masters (packed to the compute dtype by the op)
    bg = b.cuda().requires_grad_(True)
    y = ops_tfm.linear_nd(xg, wg, bg)
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    xr = x.float().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, w.float(), b)
    yr.backward(dy.float())
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-5
    assert float((y.float().cpu() - yr).abs().max()) <= tol * float(yr.abs().max()), mkn
    assert float((xg.grad.float().cpu() - xr.grad).abs().max()) <= tol * float(xr.grad.abs().max()), mkn
