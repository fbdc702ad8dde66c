"""Per-kernel parity on a real MI355X: every HIP kernel, through the C-ABI / autograd Functions,
against a plain PyTorch fp32 CPU computation of the same op on identical seeded inputs.

Tolerances (scale-relative, conftest.rel_err):
  fp32 parity mode : 1e-4  (exact-f32 MFMA, fp32 statistics; only summation order differs)
  bf16 perf mode   : 2e-2  (inputs rounded to bf16 BEFORE the reference op, so the remaining
                            error is bf16 rounding of the stored outputs, 2^-8 relative)
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def _ops():
    from simpleaicv_pytorch_training_examples_amd import ops
    return ops


def _q(t, dt):
    """round a CPU fp32 tensor through the compute dtype"""
    return t.to(dt).float()


def _nhwc_dev(t, dt):
    return t.to(dt).cuda().contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # N, C, H, W, K, R, stride, pad
    (2, 64, 14, 14, 64, 3, 1, 1),
    (3, 64, 15, 13, 128, 3, 2, 1),      # odd spatial, stride 2
    (2, 128, 9, 9, 256, 1, 1, 0),
    (2, 256, 10, 10, 64, 1, 2, 0),      # 1x1 stride 2 (downsample), narrow N tile
    (2, 8, 20, 20, 64, 7, 2, 3),        # stem geometry with padded channels
    (1, 32, 7, 7, 40, 3, 1, 1),         # K not a multiple of the tile, small M
    (5, 512, 7, 7, 512, 3, 1, 1),       # deep K = 4608
]


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_bn_act_block(case, dt):
    """ConvBnActFn forward + backward (conv fwd/dgrad/wgrad + BN + residual + ReLU)."""
    ops = _ops()
    n, c, h, w, k, r, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = _q(torch.randn(n, c, h, w, generator=g), dt)
    wt = torch.randn(k, c, r, r, generator=g) * (2.0 / (c * r * r)) ** 0.5
    gamma = torch.rand(k, generator=g) + 0.5
    beta = torch.randn(k, generator=g) * 0.1
    oh = (h + 2 * pad - r) // stride + 1
    ow = (w + 2 * pad - r) // stride + 1
    res = _q(torch.randn(n, k, oh, ow, generator=g), dt)
    dz = _q(torch.randn(n, k, oh, ow, generator=g), dt)

    for use_res, relu in [(True, True), (False, True), (False, False)]:
        # ---- reference (CPU fp32; weights rounded like the kernel sees them)
        xr = x.clone().requires_grad_(True)
        wr = wt.clone().requires_grad_(True)
        gr = gamma.clone().requires_grad_(True)
        br = beta.clone().requires_grad_(True)
        rr = res.clone().requires_grad_(True)
        y = F.conv2d(xr, _q(wr, dt) if dt == torch.bfloat16 else wr, None, stride, pad)
        if dt == torch.bfloat16:
            y = y + (_q(y.detach(), dt) - y.detach())        # straight-through bf16 storage of y
        rm, rv = torch.zeros(k), torch.ones(k)
        z = F.batch_norm(y, rm, rv, gr, br, True, 0.1, 1e-5)
        if use_res:
            z = z + rr
        if relu:
            z = F.relu(z)
        z.backward(dz)

        # ---- device
        bn = torch.nn.BatchNorm2d(k).cuda()
        with torch.no_grad():
            bn.weight.copy_(gamma)
            bn.bias.copy_(beta)
        xd = _nhwc_dev(x, dt).requires_grad_(True)
        wd = wt.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        rd = _nhwc_dev(res, dt).requires_grad_(True) if use_res else None
        zd = ops.conv_bn_act(xd, wd, bn, stride, pad, relu, rd)
        zd.backward(_nhwc_dev(dz, dt))
        torch.cuda.synchronize()
        tol = TOL[dt]
        tag = f'{case} res={use_res} relu={relu}'
        assert rel_err(zd.float(), z) < tol, 'z ' + tag
        assert rel_err(bn.running_mean, rm) < max(tol, 1e-4), 'running_mean ' + tag
        assert rel_err(bn.running_var, rv) < max(tol, 1e-4), 'running_var ' + tag
        assert int(bn.num_batches_tracked) == 1
        # gradients: the BN backward amplifies storage rounding; compare on the gradient scale
        gtol = tol * 4
        assert rel_err(xd.grad.float(), xr.grad) < gtol, 'dx ' + tag
        assert rel_err(wd.grad, wr.grad) < gtol, 'dw ' + tag
        assert rel_err(bn.weight.grad, gr.grad) < gtol, 'dgamma ' + tag
        assert rel_err(bn.bias.grad, br.grad) < gtol, 'dbeta ' + tag
        if use_res:
            assert rel_err(rd.grad.float(), rr.grad) < gtol, 'dres ' + tag


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_conv_bn_eval_mode(dt):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = _q(torch.randn(2, 64, 12, 12, generator=g), dt)
    wt = torch.randn(128, 64, 3, 3, generator=g) * 0.05
    bn = torch.nn.BatchNorm2d(128)
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(128, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(128, generator=g) + 0.5)
        bn.weight.copy_(torch.rand(128, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(128, generator=g) * 0.1)
    bn.eval()
    ref = F.relu(bn(F.conv2d(x, _q(wt, dt) if dt == torch.bfloat16 else wt, None, 1, 1)))
    bnd = torch.nn.BatchNorm2d(128)
    bnd.load_state_dict(bn.state_dict())
    bnd = bnd.cuda().eval()
    with torch.no_grad():
        out = ops.conv_bn_act(_nhwc_dev(x, dt), wt.cuda(), bnd, 1, 1, True, None)
    assert rel_err(out.float(), ref) < TOL[dt]
    assert int(bnd.num_batches_tracked) == 0


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(4, 512, 1000), (256, 2048, 1000), (7, 64, 12), (130, 200, 136)])
def test_linear(shape, dt):
    ops = _ops()
    b, ci, co = shape
    g = torch.Generator().manual_seed(b + ci)
    x = _q(torch.randn(b, ci, generator=g), dt)
    wt = torch.randn(co, ci, generator=g) * ci ** -0.5
    bias = torch.randn(co, generator=g)
    dy = torch.randn(b, co, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yr = F.linear(xr, _q(wr, dt) if dt == torch.bfloat16 else wr, br)
    yr.backward(_q(dy, dt))
    xd = x.to(dt).cuda().requires_grad_(True)
    wd = wt.cuda().requires_grad_(True)
    bd = bias.cuda().requires_grad_(True)
    yd = ops.linear(xd, wd, bd, out_f32=True)
    assert yd.dtype == torch.float32
    yd.backward(dy.cuda())
    tol = TOL[dt]
    assert rel_err(yd, yr) < tol
    assert rel_err(xd.grad.float(), xr.grad) < tol * 2
    assert rel_err(wd.grad, wr.grad) < tol * 2
    assert rel_err(bd.grad, br.grad) < tol * 2


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 64, 16, 16), (3, 64, 15, 17), (1, 8, 5, 5)])
def test_maxpool(shape, dt):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    x = _q(F.relu(torch.randn(*shape, generator=g)), dt)     # post-ReLU zeros -> many ties
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = _q(torch.randn(yr.shape, generator=g), dt)
    yr.backward(dy)
    xd = _nhwc_dev(x, dt).requires_grad_(True)
    yd = ops.max_pool2d(xd, 3, 2, 1)
    yd.backward(_nhwc_dev(dy, dt))
    assert rel_err(yd.float(), yr) == 0.0
    assert rel_err(xd.grad.float(), xr.grad) < TOL[dt]      # tie-break rule = ATen's first max


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_global_avgpool(dt):
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    x = _q(torch.randn(3, 256, 7, 7, generator=g), dt)
    xr = x.clone().requires_grad_(True)
    yr = F.adaptive_avg_pool2d(xr, (1, 1)).flatten(1)
    dy = _q(torch.randn(3, 256, generator=g), dt)
    yr.backward(dy)
    xd = _nhwc_dev(x, dt).requires_grad_(True)
    yd = ops.global_avg_pool(xd)
    yd.backward(dy.to(dt).cuda())
    assert rel_err(yd.float(), yr) < TOL[dt]
    assert rel_err(xd.grad.float(), xr.grad) < TOL[dt]


@pytest.mark.parametrize('b,c', [(8, 100), (256, 1000), (3, 7)])
def test_softmax_ce(b, c):
    ops = _ops()
    g = torch.Generator().manual_seed(b * c)
    logits = torch.randn(b, c, generator=g) * 3
    label = torch.randint(0, c, (b,), generator=g)
    soft = torch.softmax(torch.randn(b, c, generator=g), -1)
    for is_soft, lab in [(False, label), (True, soft)]:
        lr_ = logits.clone().requires_grad_(True)
        if is_soft:
            ref = torch.sum(-lab * F.log_softmax(lr_, -1), -1).mean()
        else:
            ref = F.cross_entropy(lr_, lab)
        (ref * 7.0).backward()
        ld = logits.cuda().requires_grad_(True)
        out = ops.softmax_cross_entropy(ld, lab.cuda(), soft=is_soft)
        (out * 7.0).backward()
        assert abs(float(out) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
        assert rel_err(ld.grad, lr_.grad) < 1e-5


def test_pack_input_layouts():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    nhwc = torch.randn(2, 9, 11, 3, generator=g)
    for src in (nhwc.permute(0, 3, 1, 2), nhwc.permute(0, 3, 1, 2).contiguous()):   # NHWC-strided and true NCHW
        for dt in (torch.float32, torch.bfloat16):
            out = ops.pack_input(src.cuda(), dt)
            assert out.shape == (2, 8, 9, 11) and out.is_contiguous(memory_format=torch.channels_last)
            assert torch.equal(out[:, :3].float().cpu(), src.to(dt).float())
            assert float(out[:, 3:].abs().sum()) == 0.0


def test_product_path_has_no_cpu_fallback():
    ops = _ops()
    with pytest.raises(RuntimeError):
        ops.pack_input(torch.randn(1, 3, 8, 8), torch.float32)


# ------------------------------------------------------------------------------ transformer kernels
def _tfm():
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    return ops_tfm


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(3, 17, 192), (2, 197, 768), (5, 64)])
def test_layernorm(shape, dt):
    T = _tfm()
    g = torch.Generator().manual_seed(21)
    c = shape[-1]
    x = _q(torch.randn(*shape, generator=g) * 2 + 0.3, dt)
    w = torch.rand(c, generator=g) + 0.5
    b = torch.randn(c, generator=g) * 0.1
    dy = _q(torch.randn(*shape, generator=g), dt)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (c,), wr, br, 1e-6)
    yr.backward(dy)
    xd = x.to(dt).cuda().requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yd = T.layer_norm(xd, wd, bd, 1e-6)
    yd.backward(dy.to(dt).cuda())
    tol = TOL[dt]
    assert rel_err(yd.float(), yr) < tol
    assert rel_err(xd.grad.float(), xr.grad) < tol * 2
    assert rel_err(wd.grad, wr.grad) < tol * 2
    assert rel_err(bd.grad, br.grad) < tol * 2


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gelu(dt):
    T = _tfm()
    g = torch.Generator().manual_seed(22)
    x = _q(torch.randn(7, 33, 64, generator=g) * 2, dt)
    dy = _q(torch.randn(7, 33, 64, generator=g), dt)
    xr = x.clone().requires_grad_(True)
    yr = F.gelu(xr)
    yr.backward(dy)
    xd = x.to(dt).cuda().requires_grad_(True)
    yd = T.gelu(xd)
    yd.backward(dy.to(dt).cuda())
    assert rel_err(yd.float(), yr) < TOL[dt]
    assert rel_err(xd.grad.float(), xr.grad) < TOL[dt]


def _ref_attention(qkv, heads, scale):
    b, n, c3 = qkv.shape
    c = c3 // 3
    t = qkv.view(b, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    attn = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(b, n, c)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('b,n,heads', [(2, 197, 12), (3, 17, 3), (1, 256, 2), (2, 196, 4), (1, 1, 1), (2, 33, 2)])
def test_attention(b, n, heads, dt):
    """fused attention fwd/bwd vs the reference formulation (vit.py:61-80): q k^T * scale ->
    softmax -> @ v, including the ragged last key/query tiles (n % 32 != 0) and n = 1."""
    T = _tfm()
    g = torch.Generator().manual_seed(b * 1000 + n)
    c = heads * 64
    qkv = _q(torch.randn(b, n, 3 * c, generator=g), dt)
    dy = _q(torch.randn(b, n, c, generator=g), dt)
    scale = 64 ** -0.5
    qr = qkv.clone().requires_grad_(True)
    yr = _ref_attention(qr, heads, scale)
    yr.backward(dy)
    qd = qkv.to(dt).cuda().requires_grad_(True)
    yd = T.attention(qd, heads, scale)
    yd.backward(dy.to(dt).cuda())
    tol = TOL[dt]
    assert rel_err(yd.float(), yr) < tol
    assert rel_err(qd.grad.float(), qr.grad) < tol * 2


@pytest.mark.parametrize('mode', ['1', '2'])
@pytest.mark.parametrize('b,n,heads', [(2, 197, 12), (3, 17, 3), (1, 256, 2), (2, 196, 4), (1, 1, 1), (2, 33, 2)])
def test_attention_backward_variants(b, n, heads, mode, monkeypatch):
    """SAICV_ATTN_BWD2 = 1 / 2 (two tiles / one tile per wavefront with the lean instruction mix) against the same reference
    gradient as the default backward, same tolerance; the switch is read per call."""
    T = _tfm()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(b * 1000 + n)
    c = heads * 64
    qkv = _q(torch.randn(b, n, 3 * c, generator=g), dt)
    dy = _q(torch.randn(b, n, c, generator=g), dt)
    scale = 64 ** -0.5
    qr = qkv.clone().requires_grad_(True)
    _ref_attention(qr, heads, scale).backward(dy)
    monkeypatch.setenv('SAICV_ATTN_BWD2', mode)
    qd = qkv.to(dt).cuda().requires_grad_(True)
    T.attention(qd, heads, scale).backward(dy.to(dt).cuda())
    torch.cuda.synchronize()
    assert rel_err(qd.grad.float(), qr.grad) < TOL[dt] * 2


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_attention_softmax_extremes(dt):
    """one key dominates one query by a huge margin: the max-subtracted softmax must stay finite"""
    T = _tfm()
    g = torch.Generator().manual_seed(9)
    b, n, heads = 1, 50, 1
    qkv = torch.randn(b, n, 192, generator=g)
    qkv[0, 7, 0:64] = 30.0
    qkv[0, 41, 64:128] = 30.0
    qkv = _q(qkv, dt)
    ref = _ref_attention(qkv, heads, 0.125)
    out = T.attention(qkv.to(dt).cuda(), heads, 0.125)
    assert torch.isfinite(out).all()
    assert rel_err(out.float(), ref) < TOL[dt]


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('use_drop', [False, True])
def test_vit_sublayers_match_composition(dt, use_drop):
    """fused AttnSubLayerFn / MlpSubLayerFn == x + s * f(LN(x)) composed from torch ops on CPU"""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones import vit
    T = _tfm()
    torch.manual_seed(4)
    blk = vit.TransformerEncoderLayer(192, 3, feedforward_ratio=4, drop_path_prob=0.)
    for p in blk.parameters():
        if p.ndim == 1:
            torch.nn.init.normal_(p, std=0.3)
    g = torch.Generator().manual_seed(5)
    b, n, c = 4, 17, 192
    x = _q(torch.randn(b, n, c, generator=g), dt)
    dy = _q(torch.randn(b, n, c, generator=g), dt)
    s = torch.tensor([0., 1.25, 1.25, 0.]) if use_drop else None
    W = (lambda p: _q(p, dt)) if dt == torch.bfloat16 else (lambda p: p)

    def ref_block(xr, params):
        sc = s.view(b, 1, 1) if s is not None else 1.0
        h = F.layer_norm(xr, (c,), params['norm1.weight'], params['norm1.bias'], 1e-6)
        qkv = F.linear(h, W(params['attn.qkv.weight']), params['attn.qkv.bias'])
        a = _ref_attention(qkv, 3, 64 ** -0.5)
        x1 = xr + sc * F.linear(a, W(params['attn.proj.weight']), params['attn.proj.bias'])
        h = F.layer_norm(x1, (c,), params['norm2.weight'], params['norm2.bias'], 1e-6)
        f = F.gelu(F.linear(h, W(params['mlp.fc1.weight']), params['mlp.fc1.bias']))
        return x1 + sc * F.linear(f, W(params['mlp.fc2.weight']), params['mlp.fc2.bias'])

    params = {k: v.detach().clone().requires_grad_(True) for k, v in blk.named_parameters()}
    xr = x.clone().requires_grad_(True)
    yr = ref_block(xr, params)
    yr.backward(dy)

    blk = blk.cuda()
    xd = x.to(dt).cuda().requires_grad_(True)
    sd = s.cuda() if s is not None else None
    y1 = T.attn_sublayer(xd, blk.norm1, blk.attn, sd)
    yd = T.mlp_sublayer(y1, blk.norm2, blk.mlp, sd)
    yd.backward(dy.to(dt).cuda())
    tol = TOL[dt] * (1 if dt == torch.float32 else 2)
    assert rel_err(yd.float(), yr) < tol
    assert rel_err(xd.grad.float(), xr.grad) < tol * 3
    for k, p in blk.named_parameters():
        assert rel_err(p.grad, params[k].grad) < tol * 4, k


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_gelu_linear_epilogues_equal_the_unfused_kernels(dtype):
    """saicv_linear_gelu_fwd / saicv_linear_dgrad_gelu against linear + gelu kernels run separately: the fused
    epilogues apply the activation to the same rounded values, so the results are bit-identical."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    g = torch.Generator().manual_seed(11)
    m, k, o = 300, 96, 256
    x = torch.randn(m, k, generator=g).cuda().to(dtype)
    w = (torch.randn(o, k, generator=g) * 0.2).cuda().requires_grad_(True)
    b = torch.randn(o, generator=g).cuda().requires_grad_(True)
    pre, act = ops_tfm.lin_gelu_fwd(x, w, b)
    pre_ref = ops_tfm.lin_fwd(x, w, b)
    assert torch.equal(pre, pre_ref) and torch.equal(act, ops_tfm.gelu_fwd(pre_ref))
    ref = torch.nn.functional.gelu(torch.nn.functional.linear(x.float(), w.detach(), b.detach()))
    assert rel_err(act, ref) < (2e-5 if dtype == torch.float32 else 2e-2)
    # backward of  y = act W2^T : d pre = (dy W2) * gelu'(pre)
    w2 = (torch.randn(64, o, generator=g) * 0.2).cuda().requires_grad_(True)
    dy = torch.randn(m, 64, generator=g).cuda().to(dtype)
    dpre, _, _ = ops_tfm.lin_bwd(act, w2, None, dy, gelu_pre=pre)
    dact, _, _ = ops_tfm.lin_bwd(act, w2, None, dy)
    assert torch.equal(dpre, ops_tfm.gelu_bwd(dact, pre))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_stored_gelu_derivative_form_matches_the_recomputing_form(dtype):
    """saicv_linear_gelu_fwd_aux stores gelu'(pre) where saicv_linear_gelu_fwd stores pre; saicv_linear_dgrad_mul multiplies by
    it where saicv_linear_dgrad_gelu recomputes gelu'(pre).  Same activation output bit for bit; the stored derivative equals
    gelu'(pre) of the rounded pre-activation rounded once more to the compute dtype (fp32: identical; bf16: 2^-9), and so does
    the data gradient."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    if not ops_tfm.GELU_AUX:
        pytest.skip('SAICV_GELU_AUX=0')
    g = torch.Generator().manual_seed(12)
    m, k, o = 300, 96, 256
    x = torch.randn(m, k, generator=g).cuda().to(dtype)
    w = (torch.randn(o, k, generator=g) * 0.2).cuda().requires_grad_(True)
    b = torch.randn(o, generator=g).cuda().requires_grad_(True)
    pre, act = ops_tfm.lin_gelu_fwd(x, w, b)
    dact_, act2, is_dact = ops_tfm.lin_gelu_fwd(x, w, b, aux=True)
    assert is_dact and torch.equal(act, act2)
    pf = pre.float().cpu().double()
    cdf = 0.5 * (1 + torch.erf(pf / 2 ** 0.5))
    pdf = torch.exp(-pf * pf / 2) / (2 * torch.pi) ** 0.5
    assert rel_err(dact_.float().cpu(), (cdf + pf * pdf).float()) < (1e-6 if dtype == torch.float32 else 4e-3)
    w2 = (torch.randn(64, o, generator=g) * 0.2).cuda().requires_grad_(True)
    dy = torch.randn(m, 64, generator=g).cuda().to(dtype)
    d_rec, _, _ = ops_tfm.lin_bwd(act, w2, None, dy, gelu_pre=pre)
    d_mul, _, _ = ops_tfm.lin_bwd(act, w2, None, dy, gelu_dact=dact_)
    assert rel_err(d_mul.float(), d_rec.float()) < (1e-6 if dtype == torch.float32 else 8e-3)


FUSED_DGRAD_CASES = [
    # N, C, H, W, K, R, stride, pad   (C = channels of dx = the previous BatchNorm's channels)
    (2, 64, 14, 14, 64, 3, 1, 1),
    (3, 64, 15, 13, 128, 3, 2, 1),      # stride 2: four parity classes of different sizes, zero-filled partial rows
    (2, 256, 9, 9, 64, 1, 1, 0),        # bottleneck conv1: 1x1, wide dx (two 128-column tiles)
    (9, 128, 28, 28, 128, 3, 1, 1),     # several 256-row tiles
]


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', FUSED_DGRAD_CASES)
def test_dgrad_epilogue_gated_shortcut_and_bn_backward_sums(case, dt):
    """saicv_conv2d_dgrad_fused: dx = dgrad(dy) + addend * gate bits, plus the BatchNorm-backward partial sums of dx
    (sum g, sum g * (y - mean) * invstd with g = dx * mask bits), against an fp32 CPU computation; and
    saicv_bn_act_bwd_from_partials against saicv_bn_act_bwd on the same dz."""
    import ctypes
    from simpleaicv_pytorch_training_examples_amd import _lib, ops
    from simpleaicv_pytorch_training_examples_amd._lib import check, lib, ptr
    n, c, h, w, k, r, stride, pad = case
    L, st = lib(), _lib.stream()
    epc = _lib.epc(dt)
    g = torch.Generator().manual_seed(sum(case))
    d = ops._desc(n, h, w, c, k, r, r, stride, pad, dt)
    oh, ow = d.OH, d.OW
    wgt = _q(torch.randn(k, c, r, r, generator=g) * (2.0 / (c * r * r)) ** 0.5, dt)
    dy = _q(torch.randn(n, k, oh, ow, generator=g), dt)
    addend = _q(torch.randn(n, c, h, w, generator=g), dt)
    y_bn = _q(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.7, dt)
    mean = torch.randn(c, generator=g) * 0.5 + 0.7
    invstd = torch.rand(c, generator=g) + 0.5
    gamma = torch.rand(c, generator=g) + 0.5
    M = n * h * w
    gate_bits = torch.rand(M, c, generator=g) > 0.4            # [pixel][channel], NHWC order
    mask_bits = torch.rand(M, c, generator=g) > 0.5

    def pack(bits):                                            # one byte per 16-byte chunk, bit j = element j of the chunk
        b = bits.view(M, c // epc, epc).to(torch.int32)
        return (b << torch.arange(epc, dtype=torch.int32)).sum(-1).to(torch.uint8).contiguous()

    # ---- CPU fp32 reference
    xr = torch.zeros(n, c, h, w, requires_grad=True)
    F.conv2d(xr, wgt, None, stride, pad).backward(dy)
    gate_nchw = gate_bits.view(n, h, w, c).permute(0, 3, 1, 2)
    dx_ref = xr.grad + addend * gate_nchw

    # ---- device
    wd = wgt.permute(1, 2, 3, 0).contiguous().to(dt).cuda()     # [Cin][R][S][Cout]
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    add_d = addend.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    y_d = y_bn.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    gate_d, mask_d = pack(gate_bits).cuda(), pack(mask_bits).cuda()
    mean_d, invstd_d, gamma_d = mean.cuda(), invstd.cuda(), gamma.cuda()
    rows = L.saicv_conv2d_dgrad_stat_rows(ctypes.byref(d))
    part = torch.full((2, rows, c), float('nan'), device='cuda')
    dx = torch.empty(n, h, w, c, dtype=dt, device='cuda')
    f = _lib.DgradFuse()
    f.addend, f.addend_gate = ptr(add_d), ptr(gate_d)
    f.bn_y, f.bn_mask, f.bn_mean, f.bn_invstd = ptr(y_d), ptr(mask_d), ptr(mean_d), ptr(invstd_d)
    f.part_g, f.part_gx = ptr(part[0]), ptr(part[1])
    check(L.saicv_conv2d_dgrad_fused(ctypes.byref(d), ptr(dyd), ptr(wd), ctypes.byref(f), ptr(dx), st), 'dgrad_fused')
    torch.cuda.synchronize()
    assert rel_err(dx.permute(0, 3, 1, 2).float(), dx_ref) < TOL[dt]
    # the sums are over what was STORED (dx rounded to dt), so they are checked against the device dx itself
    gd = dx.float().cpu().view(M, c) * mask_bits
    sg_ref = gd.double().sum(0)
    sgx_ref = (gd.double() * (y_bn.permute(0, 2, 3, 1).reshape(M, c).double() - mean.double()) * invstd.double()).sum(0)
    assert torch.isfinite(part).all()
    assert rel_err(part[0].double().sum(0), sg_ref) < 1e-4
    assert rel_err(part[1].double().sum(0), sgx_ref) < 1e-4

    # ---- BatchNorm backward from those partial sums == the three-pass kernel on the same dz
    ws = torch.empty(L.saicv_bn_bwd_ws_floats(M, c, _lib.dtype_code(dt)), device='cuda')
    out = []
    for fused in (False, True):
        dyb = torch.empty_like(dx)
        dgam, dbet = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
        if fused:
            check(L.saicv_bn_act_bwd_from_partials(_lib.dtype_code(dt), ptr(dx), ptr(mask_d), ptr(y_d), ptr(gamma_d), ptr(mean_d),
                                                   ptr(invstd_d), ptr(part[0]), ptr(part[1]), rows, ptr(dyb), 0, ptr(dgam),
                                                   ptr(dbet), M, c, 1, 0, ptr(ws), st), 'bn_from_partials')
        else:
            check(L.saicv_bn_act_bwd(_lib.dtype_code(dt), ptr(dx), 0, ptr(mask_d), ptr(y_d), ptr(gamma_d), ptr(mean_d), ptr(invstd_d),
                                     ptr(dyb), 0, ptr(dgam), ptr(dbet), M, c, 1, 0, ptr(ws), st), 'bn_bwd')
        torch.cuda.synchronize()
        out.append((dyb.float(), dgam.clone(), dbet.clone()))
    assert rel_err(out[1][1], out[0][1]) < 1e-4 and rel_err(out[1][2], out[0][2]) < 1e-4
    assert rel_err(out[1][0], out[0][0]) < (1e-4 if dt == torch.float32 else 8e-3)     # a flipped bf16 rounding at most


def _two_layer_grads(fuse, branch):
    """conv-bn-relu -> conv-bn-relu (+ residual), optionally with a second consumer of the first block's output."""
    import torch.nn as nn
    ops = _ops()
    ops.BN_FUSE = fuse
    torch.manual_seed(5)
    c = 32
    w1 = (torch.randn(c, c, 3, 3) * 0.1).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w2 = (torch.randn(c, c, 3, 3) * 0.1).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bn1, bn2 = nn.BatchNorm2d(c).cuda(), nn.BatchNorm2d(c).cuda()
    x = torch.randn(4, c, 12, 12).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    z1 = ops.conv_bn_act(x, w1, bn1, 1, 1, True)
    out, skip = ops.conv_bn_act(z1, w2, bn2, 1, 1, True, want_skip=True)       # skip aliases z1
    z2 = ops.conv_bn_act(out, w1, bn1, 1, 1, True, residual=skip)
    loss = (z2.float() ** 2).sum()
    if branch:
        loss = loss + (z1.float() * 0.5).sum()          # a second consumer of z1: autograd sums two gradients for it
    loss.backward()
    torch.cuda.synchronize()
    return [t.grad.float().clone() for t in (x, w1, w2, bn1.weight, bn1.bias, bn2.weight, bn2.bias)]


@pytest.mark.parametrize('branch', [False, True])
def test_bn_backward_fusions_equal_the_three_pass_form(branch):
    """SAICV_BN_FUSE paths (gated shortcut gradient, reduction in the data-gradient epilogue) against the three-pass
    BatchNorm backward on a small residual chain in fp32 -- also when a block output has a second consumer, where
    autograd hands the BatchNorm node a SUM in another tensor and the fused reduction must not be used."""
    ops = _ops()
    old = ops.BN_FUSE
    try:
        ref = _two_layer_grads(False, branch)
        got = _two_layer_grads(True, branch)
    finally:
        ops.BN_FUSE = old
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 1e-4


def test_gated_shortcut_gradient_fails_loudly_when_its_tensor_has_another_consumer():
    """The shortcut alias used twice: autograd accumulates into (or replaces) the gated gradient before the node that
    applies the gate sees it.  That must be an error, never a silently unmasked gradient."""
    import torch.nn as nn
    ops = _ops()
    if not ops.BN_FUSE:
        pytest.skip('SAICV_BN_FUSE=0')
    c = 32
    w = (torch.randn(c, c, 3, 3) * 0.1).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bn = nn.BatchNorm2d(c).cuda()
    x = torch.randn(2, c, 8, 8).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out, skip = ops.conv_bn_act(x, w, bn, 1, 1, True, want_skip=True)
    z = ops.conv_bn_act(out, w, bn, 1, 1, True, residual=skip)
    with pytest.raises(RuntimeError, match='SAICV_BN_FUSE=0'):
        ((z.float() ** 2).sum() + (skip.float() ** 2).sum()).backward()
    ops._GateLedger.pending, ops._GateLedger.queued = 0, False


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('hw', [(32, 32), (37, 45)])
def test_stem_on_the_space_to_depth_image_equals_the_strided_convolution(hw, dt):
    """7x7 stride-2 padding-3 stem (reference resnet.py:172-174) through saicv_pack_input_s2d / saicv_pack_weight_s2d /
    saicv_unpack_wgrad_s2d against F.conv2d + batch_norm + relu on the CPU in fp32; odd image sizes included."""
    import torch.nn as nn
    ops = _ops()
    h, w = hw
    torch.manual_seed(h * w)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    bn = nn.BatchNorm2d(64)
    x = torch.randn(3, h, w, 3).permute(0, 3, 1, 2)              # NHWC memory, as the collater hands it over
    wq = _q(conv.weight.detach(), dt)
    xq = _q(x, dt)
    ref_w = wq.clone().requires_grad_(True)
    y_ref = F.relu(F.batch_norm(F.conv2d(xq, ref_w, None, 2, 3), None, None, bn.weight.detach(), bn.bias.detach(), True, 0.1, bn.eps))
    g = torch.randn_like(y_ref)
    y_ref.backward(g)
    conv_d, bn_d = nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda(), nn.BatchNorm2d(64).cuda()
    with torch.no_grad():
        conv_d.weight.copy_(wq)
    conv_d.weight.data = conv_d.weight.data.contiguous(memory_format=torch.channels_last)
    assert ops.STEM_S2D
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
        xp = ops.pack_stem_input(x.cuda(), conv_d, dt)
        assert getattr(xp, '_saicv_s2d', None) is not None and xp.shape[1] == 16
        y = ops.conv_bn_act(xp, conv_d.weight, bn_d, 2, 3, True)
    y.backward(g.cuda().to(y.dtype).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    assert rel_err(y.float(), y_ref) < TOL[dt]
    assert rel_err(conv_d.weight.grad, ref_w.grad) < (1e-3 if dt == torch.float32 else 3e-2)
    assert rel_err(bn_d.running_var, torch.ones(64) * 0.9 + 0.1 * F.conv2d(xq, wq, None, 2, 3).transpose(0, 1).flatten(1).var(1)) < (1e-3 if dt == torch.float32 else 2e-2)
