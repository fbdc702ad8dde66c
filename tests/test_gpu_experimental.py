"""Parity tests of kernels that are OFF by default and waiting for their first GPU run (prepared at the end of round 4 without GPU
budget).  Skipped unless SAICV_TEST_EXPERIMENTAL=1, so that the default `pytest -m gpu` suite only holds measured code:

    SAICV_TEST_EXPERIMENTAL=1 python -m pytest tests/test_gpu_experimental.py -q -s
"""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('SAICV_TEST_EXPERIMENTAL') != '1', reason='experimental kernels: set SAICV_TEST_EXPERIMENTAL=1')]


@pytest.mark.parametrize('rows', [1, 2, 7, 16, 197 * 8 + 1, 50432])
@pytest.mark.parametrize('with_addend', [False, True])
def test_layernorm_two_rows_per_wavefront_matches_the_default_kernels(rows, with_addend):
    """SAICV_LN_HALF=1 (layernorm_{fwd,bwd}_half_kernel, C = 768 bf16): the same y / mean / rstd / dx / dgamma / dbeta as the one-row
    kernels up to fp32 summation order, and both against torch's fp32 LayerNorm; odd row counts leave the last wavefront half empty."""
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
    """SAICV_TN_INCR=1 (igemm_tn_dma_kernel<..., INCR>): the weight gradient of a bf16 convolution equals the default kernel's up to the
    order of the fp32 atomics, and both equal torch's fp32 convolution gradient at bf16 resolution."""
    from simpleaicv_pytorch_training_examples_amd import ops
    n, ci, h, w, co, k, stride, pad = geom
    g = torch.Generator().manual_seed(sum(geom))
    x = torch.randn(n, h, w, ci, generator=g).permute(0, 3, 1, 2).cuda().bfloat16()
    wt = (torch.randn(co, ci, k, k, generator=g) * 0.1)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, oh, ow, co, generator=g).permute(0, 3, 1, 2).cuda().bfloat16()
    grads = {}
    for flag in ('0', '1'):
        os.environ['SAICV_TN_INCR'] = flag
        try:
            wp = wt.clone().cuda().requires_grad_(True)
            ops.bump_weights_epoch()
            y = ops.conv2d(x.clone().requires_grad_(True), wp, None, stride, pad)
            y.backward(dy)
            torch.cuda.synchronize()
        finally:
            os.environ.pop('SAICV_TN_INCR', None)
        grads[flag] = wp.grad.float().cpu()
    ref = torch.nn.grad.conv2d_weight(x.float().cpu(), wt.shape, dy.float().cpu(), stride=stride, padding=pad)
    scale = float(ref.abs().max())
    assert float((grads['0'] - grads['1']).abs().max()) <= 1e-4 * scale, geom
    assert float((grads['1'] - ref).abs().max()) <= 2e-2 * scale, geom
