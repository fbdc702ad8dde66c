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
                                 (50432, 768, 2304), (5000, 3072, 768), (50432, 768, 768)])
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


def test_fused_residual_and_drop_path_rows_at_vit_size():
    """A ViT-B projection at the bench size (M = 197 * 256 rows, N = K = 768: 1 182 tiles of 256 x 128) with the residual addend and the
    drop-path factor (one per group of 197 rows) fused into the epilogue, against torch fp32.  Written for the split launch r05 tried
    and dropped (csrc/igemm.hip above igemm_nt); kept because no other test runs the row-group factor at a size where groups
    straddle tile rows: rows around tile-row and group boundaries are checked one by one."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    m, k, n = 197 * 256, 768, 768
    g = torch.Generator().manual_seed(11)
    x = torch.randn(m, k, generator=g).bfloat16()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().float()
    b = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g).bfloat16()
    scale = ((torch.rand(m // 197, generator=g) > 0.3).float() / 0.7)
    y = ops_tfm.lin_fwd(x.cuda(), w.cuda(), b.cuda(), addend=res.cuda(), row_scale=scale.cuda(), rows_per_scale=197)
    torch.cuda.synchronize()
    ref = res.float() + scale.repeat_interleave(197)[:, None] * (x.float() @ w.t() + b)
    err = (y.float().cpu() - ref).abs()
    assert float(err.max()) <= 2e-2 * float(ref.abs().max())
    # rows on both sides of a tile-row boundary inside a group, and that group's first / last row
    for r in (0, 43519, 43520, 43521, 43520 + 6911, (43520 // 197) * 197, (43520 // 197 + 1) * 197 - 1):
        assert float(err[r].max()) <= 2e-2 * float(ref.abs().max()), r


def _scipy_pairs(cost, valid):
    import numpy as np
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    crit = DETRLoss(num_classes=20)
    out = []
    for i in range(cost.shape[0]):
        cols = np.nonzero(valid[i].numpy())[0]
        if cols.size == 0:
            out.append(set())
            continue
        rows, cj = crit.linear_sum_assignment_with_inf(cost[i][:, cols].numpy())
        out.append({(int(r), int(cols[c])) for r, c in zip(rows, cj)})
    return out


@pytest.mark.parametrize('case', ['float_tall', 'float_square', 'float_wide', 'integer_ties', 'constant', 'nan_inf', 'empty_and_full'])
def test_device_hungarian_assignment_equals_scipy(case):
    """saicv_detr_assign (csrc/detloss.hip: scipy's rectangular LSAP restated, one workgroup per image) against
    scipy.optimize.linear_sum_assignment through the reference's nan / inf wrapper (reference SimpleAICV/detection/losses.py:1063-1090,
    DETRLoss.linear_sum_assignment_with_inf): the SAME pairs, not just the same total -- integer costs full of ties and a constant
    matrix check scipy's tie rules (reverse-filled `remaining` list, swap removal, ties towards a new sink); fewer, as many and more
    ground-truth boxes than queries; padding rows interleaved with boxes; nan and one-signed infinities."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    g = torch.Generator().manual_seed(hash(case) % 1000)
    b, q, t = 6, 20, 32
    valid = torch.zeros(b, t, dtype=torch.bool)
    if case == 'float_tall':
        counts = [1, 3, 7, 12, 19, 5]
    elif case == 'float_square':
        counts = [20] * b
    elif case == 'float_wide':
        counts = [21, 25, 32, 28, 22, 30]
    elif case == 'empty_and_full':
        counts = [0, 32, 0, 1, 20, 19]
    else:
        counts = [4, 9, 20, 27, 13, 2]
    for i, c in enumerate(counts):
        perm = torch.randperm(t, generator=g)[:c]          # boxes interleaved with padding rows
        valid[i, perm] = True
    cost = torch.randn(b, q, t, generator=g)
    if case == 'integer_ties':
        cost = torch.randint(0, 4, (b, q, t), generator=g).float()
    elif case == 'constant':
        cost = torch.full((b, q, t), 0.25)
    elif case == 'nan_inf':
        cost[0, 3, :] = float('nan')
        cost[1, :, valid[1].nonzero()[0, 0]] = float('inf')
        cost[2, 5, valid[2].nonzero()[1, 0]] = float('-inf')
        cost[3][torch.rand(q, t, generator=g) < 0.1] = float('inf')
    crit = DETRLoss(num_classes=20)
    src, tgt, w = crit.assign_device(cost.cuda(), valid.cuda())
    torch.cuda.synchronize()
    src, tgt, w = src.cpu(), tgt.cpu(), w.cpu()
    want = _scipy_pairs(cost, valid)
    for i in range(b):
        got = {(int(s), int(k)) for s, k, ww in zip(src[i], tgt[i], w[i]) if ww > 0}
        assert int((w[i] > 0).sum()) == min(counts[i], q), (case, i)
        assert got == want[i], (case, i, sorted(got ^ want[i])[:6])


@pytest.mark.parametrize('heads', [8, 16], ids=['head_dim_64', 'head_dim_32'])
def test_detr_attention_with_shared_qk_projection_at_both_head_dims(heads):
    """detr._mha with same_qk (encoder / decoder self-attention: q = k = x + pos, v = x) against nn.MultiheadAttention itself, for
    head dimension 64 (hidden 512 / 8 heads -- ADVICE r04: K used to be a column slice of the packed [q | k] projection, whose row
    stride the streaming kernel's single K / V stride rejects) and 32 (the packed projection stays).  fp32: output 1e-4, every
    gradient (inputs, packed in-projection, out-projection) 1e-3 of its scale; an additive key-padding bias (-30) on the last keys of image 1."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import detr
    torch.manual_seed(5)
    c, b, n = 512, 2, 80
    mha = torch.nn.MultiheadAttention(c, heads, dropout=0.0).cuda()
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.1)
        mha.out_proj.bias.normal_(0, 0.1)
    x = torch.randn(b, n, c, device='cuda')
    pos = torch.randn(b, n, c, device='cuda')
    pad = torch.zeros(b, n, dtype=torch.bool, device='cuda')
    pad[1, 70:] = True
    probe = torch.randn(b, n, c, device='cuda')
    # torch's own module (sequence-first), fp32
    xr, pr = x.clone().requires_grad_(True), pos.clone()
    qk = (xr + pr).transpose(0, 1)
    key_bias = torch.zeros(b, n, device='cuda').masked_fill(pad, -30.0)      # additive float mask, as DETR hands it over
    ref = mha(qk, qk, xr.transpose(0, 1), key_padding_mask=key_bias, need_weights=False)[0].transpose(0, 1)
    (ref * probe).sum().backward()
    ref_g = {'x': xr.grad.clone(), 'in_w': mha.in_proj_weight.grad.clone(), 'in_b': mha.in_proj_bias.grad.clone(),
             'out_w': mha.out_proj.weight.grad.clone(), 'out_b': mha.out_proj.bias.grad.clone()}
    mha.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_(True)
    q_in = xg + pos
    out = detr._mha(mha, q_in, q_in, xg, key_bias, True)
    (out.float() * probe).sum().backward()
    torch.cuda.synchronize()
    assert float((out.float() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    got = {'x': xg.grad, 'in_w': mha.in_proj_weight.grad, 'in_b': mha.in_proj_bias.grad, 'out_w': mha.out_proj.weight.grad,
           'out_b': mha.out_proj.bias.grad}
    for k, r in ref_g.items():
        assert got[k] is not None, k
        assert float((got[k].float() - r).abs().max()) <= 1e-3 * float(r.abs().max()), k
