"""Streaming attention kernels and the SAM image encoder on a real MI355X.

Kernel level : saicv_attention_stream_{fwd,bwd} (through the C-ABI) against a plain fp32 PyTorch
               restatement of softmax(scale q k^T + bias) v.  fp32 parity mode: outputs and all
               gradients within 2e-4 of the tensor scale (exact-f32 MFMA, only summation order
               differs); bf16: within 3e-2 (operands and probabilities rounded to bf16).
Model level  : ViTImageEncoder against the fixture produced by the reference's own module
               (tests/golden/sam_encoder_tiny.pt): fp32 output within 1e-3 (north_star), gradient
               norms within 1e-2, samples within 2e-2; bf16 autocast measured against the
               reference's own bf16-vs-fp32 deviation stored in the fixture.
"""
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


def _ref_attention(q, k, v, heads, scale, key_bias=None, rel_h=None, rel_w=None):
    b, nq, c = q.shape
    nk = k.shape[1]
    d = c // heads
    qh = q.view(b, nq, heads, d).transpose(1, 2)
    kh = k.view(b, nk, heads, d).transpose(1, 2)
    vh = v.view(b, nk, heads, d).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale                       # [b, h, nq, nk]
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    if rel_h is not None:
        sh, sw = rel_h.shape[-1], rel_w.shape[-1]
        s = (s.view(b, heads, nq, sh, sw) + rel_h.view(b, heads, nq, sh, 1) + rel_w.view(b, heads, nq, 1, sw)).view(
            b, heads, nq, nk)
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(b, nq, c)


CASES = [
    # b, heads, d, nq, nk, key_bias, rel (sh, sw)
    (2, 3, 64, 196, 196, False, (14, 14)),     # SAM window
    (1, 2, 64, 256, 256, False, (16, 16)),     # global block, several key chunks
    (1, 2, 64, 192, 192, False, (3, 64)),      # Sw = 64: the register path of the 64x64 global blocks
    (2, 1, 64, 320, 320, True, (5, 64)),       # same, more chunks, plus a key bias
    (1, 1, 64, 400, 400, False, (20, 20)),     # generic table path (LDS atomics)
    (2, 2, 64, 200, 200, False, None),         # ragged tails, no bias
    (2, 8, 32, 100, 330, True, None),          # DETR cross-attention: head dim 32, additive key bias
    (2, 8, 32, 330, 330, True, None),          # DETR encoder self-attention
    (1, 1, 64, 1, 70, False, None),            # single query
    (1, 2, 32, 130, 5, True, None),            # fewer keys than one tile
    (1, 2, 64, 4096, 4096, False, (64, 64)),   # BASELINE.json configs[4]: a SAM-B global block at 1024 x 1024 (64 x 64 tokens)
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', CASES)
def test_stream_attention_matches_fp32_reference(case, dtype):
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    b, heads, d, nq, nk, use_kb, rel = case
    c = heads * d
    g = torch.Generator().manual_seed(nq * 131 + nk)
    packed = nq == nk
    if packed:      # q / k / v are views of one packed projection, as in the encoder blocks
        qkv = torch.randn(b, nq, 3 * c, generator=g).cuda().to(dtype)
        q, k, v = qkv[:, :, :c], qkv[:, :, c:2 * c], qkv[:, :, 2 * c:]
    else:
        q = torch.randn(b, nq, c, generator=g).cuda().to(dtype)
        k = torch.randn(b, nk, c, generator=g).cuda().to(dtype)
        v = torch.randn(b, nk, c, generator=g).cuda().to(dtype)
    kb = (torch.rand(b, nk, generator=g) > 0.7).float().cuda() * 1.0 if use_kb else None
    rel_h = rel_w = None
    if rel:
        rel_h = torch.randn(b * heads, nq, rel[0], generator=g).cuda()
        rel_w = torch.randn(b * heads, nq, rel[1], generator=g).cuda()
    scale = d ** -0.5
    out, lse = ops_tfm.sattn_fwd(q, k, v, heads, scale, kb, rel_h, rel_w)
    dout = torch.randn(b, nq, c, generator=g).cuda().to(dtype)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    if packed:
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[:, :, :c], dqkv[:, :, c:2 * c], dqkv[:, :, 2 * c:]
    drh, drw = ops_tfm.sattn_bwd(q, k, v, out, dout, lse, heads, scale, dq, dk, dv, kb, rel_h, rel_w)
    torch.cuda.synchronize()

    leaves = [t.detach().float().clone().requires_grad_(True) for t in (q, k, v)]
    rl = [t.detach().clone().requires_grad_(True) for t in (rel_h, rel_w)] if rel else [None, None]
    ref = _ref_attention(leaves[0], leaves[1], leaves[2], heads, scale, kb, rl[0], rl[1])
    ref.backward(dout.float())
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(out, ref) < tol
    assert rel_err(dq, leaves[0].grad) < tol
    assert rel_err(dk, leaves[1].grad) < tol
    assert rel_err(dv, leaves[2].grad) < tol
    if rel:
        assert rel_err(drh, rl[0].grad) < tol
        assert rel_err(drw, rl[1].grad) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('pattern', ['ramp', 'spike', 'flat'])
def test_stream_attention_forward_deferred_rescale_branch(pattern, dtype):
    """The r04 forward moves its running maximum only when a row's maximum outgrows it by 2^8 (csrc/attn_stream.hip,
    sa_fwd2_kernel).  Random logits almost never take that branch after the first chunk, so it gets inputs that force it:
    `ramp` -- key norms grow chunk by chunk, the maximum creeps up by a few log2 units per chunk (cumulative growth must
    trigger the move, a single step must not have to); `spike` -- one key in the sixth chunk dominates one query by hundreds
    of units; `flat` -- identical logits (nothing ever grows).  Checked against the fp32 softmax with rel-pos bias and without."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    b, heads, d, n = 1, 2, 64, 512
    c = heads * d
    g = torch.Generator().manual_seed(17)
    q = torch.randn(b, n, c, generator=g)
    k = torch.randn(b, n, c, generator=g)
    v = torch.randn(b, n, c, generator=g)
    if pattern == 'ramp':
        k = k * (1.0 + 0.9 * (torch.arange(n) // 64).float())[None, :, None]
        q = q * 2.0
    elif pattern == 'spike':
        k[0, 5 * 64 + 7, :64] = q[0, 33, :64] * 6.0           # head 0: key 327 ~ 6 |q|^2 * scale for query 33
    else:
        q = torch.zeros_like(q)
    for rel in (None, (8, 64)):
        rel_h = rel_w = None
        if rel:
            rel_h = torch.randn(b * heads, n, rel[0], generator=g).cuda()
            rel_w = torch.randn(b * heads, n, rel[1], generator=g).cuda()
        qd, kd, vd = (t.cuda().to(dtype).contiguous() for t in (q, k, v))
        out, lse = ops_tfm.sattn_fwd(qd, kd, vd, heads, d ** -0.5, None, rel_h, rel_w)
        torch.cuda.synchronize()
        ref = _ref_attention(qd.float(), kd.float(), vd.float(), heads, d ** -0.5, None, rel_h, rel_w)
        qh = qd.float().view(b, n, heads, d).transpose(1, 2)
        kh = kd.float().view(b, n, heads, d).transpose(1, 2)
        sc = qh @ kh.transpose(-1, -2) * d ** -0.5
        if rel:
            sc = (sc.view(b, heads, n, rel[0], rel[1]) + rel_h.view(b, heads, n, rel[0], 1) + rel_w.view(b, heads, n, 1, rel[1])).view(b, heads, n, n)
        ref_lse = torch.logsumexp(sc, -1).view(b * heads, n)
        tol = 2e-4 if dtype == torch.float32 else 3e-2
        assert torch.isfinite(out.float()).all()
        assert rel_err(out, ref) < tol, (pattern, rel)
        assert float((lse - ref_lse).abs().max()) < (1e-3 if dtype == torch.float32 else 5e-2) * max(1.0, float(ref_lse.abs().max()) * 0.05), (pattern, rel)


def test_stream_attention_rejects_bad_arguments():
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    q = torch.randn(1, 16, 48, device='cuda')
    with pytest.raises(RuntimeError, match='head dim'):
        ops_tfm.sattn_fwd(q, q, q, 1, 1.0)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops_tfm.stream_attention(q.cpu(), q.cpu(), q.cpu(), 1, 1.0)


def _sam_model(fx):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import ViTImageEncoder
    torch.manual_seed(fx['model_seed'])
    m = ViTImageEncoder(**fx['kwargs'])
    O.sam_randomize_zero_init(m.named_parameters(), fx['model_seed'] + 100)
    return m.cuda().train()


def _sam_inputs(fx):
    g = torch.Generator().manual_seed(fx['data_seed'])
    s = fx['kwargs']['image_size']
    x = torch.randn(fx['batch'], 3, s, s, generator=g)
    probe = torch.randn(fx['output'].shape, generator=g)
    assert abs(float(x.double().sum()) - fx['input_checksum']) < 1e-6
    return x.cuda(), probe.cuda()


@pytest.mark.parametrize('name', ['sam_encoder_tiny', 'sam_b_encoder_256'])
def test_sam_encoder_fp32_matches_reference(name, deterministic):
    """sam_b_encoder_256: the encoder at sam_b's real dimensions (768 planes, 12 heads x 64, window 14, global blocks
    2/5/8/11) on a 256 x 256 image -- fixture produced by the reference's ViTImageEncoder (oracle/make_golden_sam.py)."""
    fx = load_golden(name)
    m = _sam_model(fx)
    x, probe = _sam_inputs(fx)
    out = m(x)
    assert out.shape == fx['output'].shape and out.dtype == torch.float32
    (out * probe).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(out, fx['output']) < 1e-3
    worst = 0.0
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        ref_n = fx['grad_norm'][n]
        assert abs(float(p.grad.norm()) - ref_n) <= 1e-2 * max(ref_n, 1e-6), (n, float(p.grad.norm()), ref_n)
        e = rel_err(p.grad.flatten()[:64], fx['grad_sample'][n])
        worst = max(worst, e)
        assert e < 2e-2, (n, e)
        if n in fx['grad_full']:
            assert rel_err(p.grad, fx['grad_full'][n]) < 2e-2, n
    print(f'{name} fp32: worst gradient-sample error {worst:.2e}')


@pytest.mark.parametrize('name', ['sam_encoder_tiny', 'sam_b_encoder_256'])
def test_sam_encoder_bf16_tracks_reference_autocast(name):
    fx = load_golden(name)
    m = _sam_model(fx)
    x, probe = _sam_inputs(fx)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = m(x)
    (out.float() * probe).sum().backward()
    torch.cuda.synchronize()
    noise = fx['reference_noise']
    assert rel_err(out.float(), fx['output']) < 1.5 * noise['bf16_output'] + 2e-2
    a = torch.cat([p.grad.flatten()[:64].double().cpu() for _, p in m.named_parameters()])
    b = torch.cat([fx['grad_sample'][n].double() for n, _ in m.named_parameters()])
    cos = float(a @ b / (a.norm() * b.norm()))
    assert cos > noise['bf16_grad_sample_cos'] - 0.1, cos


@pytest.mark.timeout(900)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_sam_b_blocks_at_1024_match_reference(dtype):
    """BASELINE.json configs[4] at bench resolution: one 3 x 1024 x 1024 image through patch embedding, ONE windowed block
    (64 x 64 tokens padded to 70 x 70, 25 windows of 196), ONE global block (4096 tokens, decomposed rel-pos over 64 x 64) and
    the neck, at sam_b's real dimensions -- fixture produced by the reference's ViTImageEncoder on the CPU
    (oracle/make_golden_r03.py; reference image_encoder.py:82-184, 201-239).  fp32: output 1e-3, gradient norms 1e-2,
    samples 2e-2; bf16 autocast against the reference's own bf16-vs-fp32 deviation."""
    fx = load_golden('sam_b_blocks_1024')
    m = _sam_model(fx)
    g = torch.Generator().manual_seed(fx['data_seed'])
    x = torch.randn(1, 3, 1024, 1024, generator=g)
    probe = torch.randn(fx['output_shape'], generator=g)
    assert abs(float(x.double().sum()) - fx['input_checksum']) < 1e-6
    assert abs(float(probe.double().sum()) - fx['probe_checksum']) < 1e-6
    x, probe = x.cuda(), probe.cuda()
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        out = m(x)
    assert list(out.shape) == fx['output_shape']
    (out.float() * probe).sum().backward()
    torch.cuda.synchronize()
    noise = fx['reference_noise']
    otol = 1e-3 if dtype == torch.float32 else 1.5 * noise['bf16_output'] + 2e-2
    assert rel_err(out.float()[:, :, ::2, ::2], fx['output_sub']) < otol
    assert rel_err(out.float()[:, :, 37, :], fx['output_row']) < otol * float(fx['output_sub'].abs().max() / fx['output_row'].abs().max())
    assert abs(float(out.float().norm()) - fx['output_norm']) < (1e-3 if dtype == torch.float32 else 2e-2) * fx['output_norm']
    if dtype == torch.float32:
        worst = 0.0
        for n, p in m.named_parameters():
            assert p.grad is not None, n
            ref_n = fx['grad_norm'][n]
            assert abs(float(p.grad.norm()) - ref_n) <= 1e-2 * max(ref_n, 1e-6), (n, float(p.grad.norm()), ref_n)
            e = rel_err(p.grad.flatten()[:64], fx['grad_sample'][n])
            worst = max(worst, e)
            assert e < 2e-2, (n, e)
            if n in fx['grad_full']:
                assert rel_err(p.grad, fx['grad_full'][n]) < 2e-2, n
        print(f'sam_b_blocks_1024 fp32: worst gradient-sample error {worst:.2e}')
    else:
        a = torch.cat([p.grad.flatten()[:64].double().cpu() for _, p in m.named_parameters()])
        b = torch.cat([fx['grad_sample'][n].double() for n, _ in m.named_parameters()])
        cos = float(a @ b / (a.norm() * b.norm()))
        assert cos > noise['bf16_grad_sample_cos'] - 0.1, cos


def test_sam_encoder_gradient_checkpoint_equals_plain(monkeypatch):
    monkeypatch.setenv('SAICV_ACTIVATION_CHECKPOINT', '1')      # on MI355X the flag is honoured only when memory requires it
    fx = load_golden('sam_encoder_tiny')
    kw = dict(fx['kwargs'])
    x, probe = _sam_inputs(fx)
    grads = []
    for ck in (False, True):
        fx2 = dict(fx)
        fx2['kwargs'] = dict(kw, use_gradient_checkpoint=ck)
        m = _sam_model(fx2)
        (m(x) * probe).sum().backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
    for n in grads[0]:
        assert rel_err(grads[1][n], grads[0][n]) < 1e-4, n


# ------------------------------------------------------------------------------------------ mask loss kernel
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_mask_loss_stats_kernel_matches_reference_formulas(dtype):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.losses import SAMLoss
    g = torch.Generator().manual_seed(5)
    b, m, h, w = 3, 4, 64, 96
    x = (torch.randn(b, m, h, w, generator=g) * 3).cuda().to(dtype)
    t = (torch.rand(b, 1, h, w, generator=g) > 0.6).float().cuda()
    ious = torch.rand(b, m, generator=g).cuda()
    crit = SAMLoss()
    xl = x.detach().clone().requires_grad_(True)
    f, d, i = crit.per_mask_losses(xl, ious, t)
    (f.sum() * 20 + d.sum() + i.sum()).backward()
    xr = x.detach().float().clone().requires_grad_(True)
    rf, rd, ri = O.sam_per_mask_losses(xr, t, ious)
    (rf.sum() * 20 + rd.sum() + ri.sum()).backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-5      # the kernel computes in fp32 from the stored logits
    assert rel_err(f, rf) < tol and rel_err(d, rd) < tol and rel_err(i, ri) < 1e-5
    gtol = 1e-4 if dtype == torch.float32 else 1e-2     # bf16: the gradient is rounded to bf16 on store
    assert rel_err(xl.grad, xr.grad) < gtol


# ------------------------------------------------------------------------------------------ decoder tail kernels (samtail.hip)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_hyper_product_matches_matmul(dtype):
    """masks = hyper_in @ upscaled_embedding (reference mask_decoder.py:137-140) and its gradients vs fp32 matmul."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    g = torch.Generator().manual_seed(3)
    b, t, p, c = 3, 4, 1000, 32                           # P not a multiple of the block
    x = torch.randn(b, p, c, generator=g).to(dtype).float()
    hy = torch.randn(b, t, c, generator=g).to(dtype).float()
    dout = torch.randn(b, t, p, generator=g).to(dtype).float()
    xr, hr = x.clone().requires_grad_(True), hy.clone().requires_grad_(True)
    ref = torch.matmul(hr, xr.transpose(1, 2))
    ref.backward(dout)
    xd, hd = x.cuda().to(dtype).requires_grad_(True), hy.cuda().to(dtype).requires_grad_(True)
    out = ops_tfm.hyper_product(xd, hd)
    out.backward(dout.cuda().to(dtype))
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert out.shape == (b, t, p) and rel_err(out.float(), ref) < tol
    assert rel_err(xd.grad.float(), xr.grad) < tol
    assert rel_err(hd.grad.float(), hr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('shape', [(2, 3, 16, 16), (1, 4, 9, 23), (2, 1, 1, 5)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_upsample4_matches_interpolate(shape, dtype):
    """x4 bilinear, align_corners=False (reference sam.py:155-158), forward and backward, borders included."""
    import torch.nn.functional as F
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    g = torch.Generator().manual_seed(sum(shape))
    low = torch.randn(shape, generator=g).to(dtype).float()
    dhi = torch.randn(shape[0], shape[1], 4 * shape[2], 4 * shape[3], generator=g).to(dtype).float()
    lr = low.clone().requires_grad_(True)
    ref = F.interpolate(lr, scale_factor=4, mode='bilinear')
    ref.backward(dhi)
    ld = low.cuda().to(dtype).requires_grad_(True)
    out = ops_tfm.upsample4_bilinear(ld)
    out.backward(dhi.cuda().to(dtype))
    torch.cuda.synchronize()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert out.shape == ref.shape and rel_err(out.float(), ref) < tol
    assert rel_err(ld.grad.float(), lr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_mask_loss_from_low_resolution_logits_matches_interpolate_then_loss(dtype):
    """SAMLoss statistics and gradient taken from the low-resolution logits (the x4 upsampled tensor never stored) vs
    F.interpolate followed by the reference formulas (oracle sam_per_mask_losses), incl. the gradient through the
    interpolation down to the low-resolution logits."""
    import torch.nn.functional as F
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.losses import SAMLoss
    g = torch.Generator().manual_seed(9)
    b, m, h, w = 3, 4, 24, 40
    low = (torch.randn(b, m, h, w, generator=g) * 3).to(dtype).float()
    t = (torch.rand(b, 1, 4 * h, 4 * w, generator=g) > 0.6).float()
    ious = torch.rand(b, m, generator=g)
    lr = low.clone().requires_grad_(True)
    rf, rd, ri = O.sam_per_mask_losses(F.interpolate(lr, scale_factor=4, mode='bilinear'), t, ious)
    (rf.sum() * 20 + rd.sum() + ri.sum()).backward()
    crit = SAMLoss()
    ld = low.cuda().to(dtype).requires_grad_(True)
    hi = ops_tfm.upsample4_bilinear(ld)
    hi._saicv_low = ld                                    # what SAM.forward_prompt_encoder_mask_decoder attaches
    f, d, i = crit.per_mask_losses(hi, ious.cuda(), t.cuda())
    (f.sum() * 20 + d.sum() + i.sum()).backward()
    torch.cuda.synchronize()
    assert rel_err(f, rf) < 2e-5 and rel_err(d, rd) < 2e-5 and rel_err(i, ri) < 1e-5
    assert rel_err(ld.grad.float(), lr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    # and the two routes agree: without the attribute the loss reads the materialised full-resolution tensor
    ld2 = low.cuda().to(dtype).requires_grad_(True)
    f2, d2, i2 = crit.per_mask_losses(ops_tfm.upsample4_bilinear(ld2), ious.cuda(), t.cuda())
    (f2.sum() * 20 + d2.sum() + i2.sum()).backward()
    assert rel_err(f2, rf) < (2e-5 if dtype == torch.float32 else 2e-3)
    assert rel_err(ld2.grad.float(), lr.grad) < (1e-4 if dtype == torch.float32 else 3e-2)


def test_sam_full_model_two_pass_matches_reference():
    """SAM (encoder + prompt encoder + mask decoder) + SAMLoss, two decoder passes, against the fixture
    produced by the reference modules (oracle/make_golden_sam.py: sam_case)."""
    from oracle.make_golden_sam import sam_inputs, sam_two_pass_loss
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything import sam
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation import losses
    fx = load_golden('sam_tiny_two_pass')
    torch.manual_seed(fx['model_seed'])
    model = sam.SAM(**fx['kwargs'])
    O.sam_randomize_zero_init(model.named_parameters(), fx['model_seed'] + 100)
    model = model.cuda().train()
    crit = losses.SAMLoss(alpha=0.25, gamma=2, focal_loss_weight=20, dice_loss_weight=1, iou_predict_loss_weight=1,
                          supervise_all_iou=True, mask_threshold=0.0)
    images, masks, points, boxes = sam_inputs(fx['kwargs'], fx['batch'], fx['data_seed'])
    assert abs(float(images.double().sum() + masks.double().sum() + points.double().sum()) - fx['input_checksum']) < 1e-6
    ld, total, mps, ips = sam_two_pass_loss(model, crit, images.cuda(), masks.cuda(), points.cuda(), boxes.cuda(),
                                            fx['kwargs']['image_size'])
    total.backward()
    torch.cuda.synchronize()
    for k in range(2):
        assert rel_err(mps[k][:, :, ::16, ::16], fx['mask_preds_sample'][k]) < 1e-3
        assert rel_err(torch.nn.functional.avg_pool2d(mps[k].float(), 4), fx['mask_preds_lowres'][k]) < 1e-3
        assert rel_err(ips[k], fx['iou_preds'][k]) < 1e-3
    for k, v in fx['loss'].items():
        assert abs(float(ld[k]) - v) < 1e-3 * max(abs(v), 1e-3), (k, float(ld[k]), v)
    worst = 0.0
    for n, p in model.named_parameters():
        if n in fx['no_grad_params']:
            continue
        assert p.grad is not None, n
        ref_n = fx['grad_norm'][n]
        assert abs(float(p.grad.norm()) - ref_n) <= 1e-2 * max(ref_n, 1e-6), (n, float(p.grad.norm()), ref_n)
        if ref_n > 1e-7:
            e = rel_err(p.grad.flatten()[:64], fx['grad_sample'][n])
            worst = max(worst, e)
            assert e < 2e-2, (n, e)
    print(f'sam_tiny_two_pass fp32: worst gradient-sample error {worst:.2e}')


def test_sam_full_model_bf16_tracks_reference_autocast():
    from oracle.make_golden_sam import sam_inputs, sam_two_pass_loss
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything import sam
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation import losses
    fx = load_golden('sam_tiny_two_pass')
    torch.manual_seed(fx['model_seed'])
    model = sam.SAM(**fx['kwargs'])
    O.sam_randomize_zero_init(model.named_parameters(), fx['model_seed'] + 100)
    model = model.cuda().train()
    crit = losses.SAMLoss()
    images, masks, points, boxes = sam_inputs(fx['kwargs'], fx['batch'], fx['data_seed'])
    ld, total, mps, ips = sam_two_pass_loss(model, crit, images.cuda(), masks.cuda(), points.cuda(), boxes.cuda(),
                                            fx['kwargs']['image_size'], autocast_dtype=torch.bfloat16, device_type='cuda')
    total.backward()
    torch.cuda.synchronize()
    noise = fx['reference_noise']
    assert rel_err(mps[0][:, :, ::16, ::16].float(), fx['mask_preds_sample'][0]) < 1.5 * noise['bf16_masks'] + 2e-2
    assert abs(float(total) - fx['total']) < (1.5 * noise['bf16_loss'] + 1e-2) * abs(fx['total'])
    names = [n for n, p in model.named_parameters() if p.grad is not None and n in fx['grad_sample']]
    a = torch.cat([dict(model.named_parameters())[n].grad.flatten()[:64].double().cpu() for n in names])
    b = torch.cat([fx['grad_sample'][n].double() for n in names])
    cos = float(a @ b / (a.norm() * b.norm()))
    assert cos > noise['bf16_grad_sample_cos'] - 0.1, cos


# ------------------------------------------------------------------------------------------ SAM layout / rel-pos kernels
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 16, 16, 64, 7), (1, 64, 64, 128, 14), (3, 14, 28, 32, 14)])
def test_window_kernels_match_reference_layout(shape, dtype):
    """saicv_window_partition / _unpartition against the reference's pad + view + permute (image_encoder.py:32-79)."""
    import torch.nn.functional as F
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    b, h, w, c, ws = shape
    g = torch.Generator().manual_seed(h * 7 + ws)
    x = torch.randn(b, h, w, c, generator=g).cuda().to(dtype)
    ph, pw = (ws - h % ws) % ws, (ws - w % ws) % ws
    xp = F.pad(x, (0, 0, 0, pw, 0, ph))
    hp, wp = h + ph, w + pw
    ref = xp.view(b, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)
    win, pad_hw = ops_tfm.window_partition(x, ws)
    assert pad_hw == (hp, wp) and torch.equal(win, ref)
    back = ops_tfm.window_unpartition(win, ws, pad_hw, (h, w))
    assert torch.equal(back, x)
    add = torch.randn(b, h, w, c, generator=g).cuda().to(dtype)
    fused = ops_tfm.window_unpartition(win, ws, pad_hw, (h, w), addend=add)
    assert rel_err(fused, (x.float() + add.float())) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('grid', [(14, 14), (64, 64), (16, 16), (5, 64)])
def test_relpos_kernels_match_einsum(grid, dtype):
    """saicv_relpos_fwd / _bwd against get_rel_pos + the einsums of add_decomposed_rel_pos (image_encoder.py:82-144)."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    sh, sw = grid
    bw, heads, d = 2, 3, 64
    n, c = sh * sw, heads * d
    g = torch.Generator().manual_seed(sh * 100 + sw)
    qkv = torch.randn(bw, n, 3 * c, generator=g).cuda().to(dtype)
    q = qkv[:, :, :c]
    th = (torch.randn(2 * sh - 1, d, generator=g) * 0.5).cuda().requires_grad_(True)
    tw = (torch.randn(2 * sw - 1, d, generator=g) * 0.5).cuda().requires_grad_(True)
    rel_h, rel_w = ops_tfm.relpos_fwd(q, heads, sh, sw, th, tw)
    ql = q.detach().float().clone().requires_grad_(True)
    rq = ql.view(bw, sh, sw, heads, d)
    rh = th[(torch.arange(sh)[:, None] - torch.arange(sh)[None, :] + sh - 1).cuda()]
    rw = tw[(torch.arange(sw)[:, None] - torch.arange(sw)[None, :] + sw - 1).cuda()]
    ref_h = torch.einsum('bhwnc,hkc->bnhwk', rq, rh).reshape(bw * heads, n, sh)
    ref_w = torch.einsum('bhwnc,wkc->bnhwk', rq, rw).reshape(bw * heads, n, sw)
    # fp32: exact-f32 arithmetic.  bf16: the logits come off the matrix cores with the TABLES rounded to bf16 as well (what the
    # reference's autocast einsum does): 2^-9 per product against the fp32-table einsum this test builds
    ftol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel_err(rel_h, ref_h) < ftol and rel_err(rel_w, ref_w) < ftol
    drh = torch.randn(bw * heads, n, sh, generator=g).cuda()
    drw = torch.randn(bw * heads, n, sw, generator=g).cuda()
    (ref_h * drh).sum().backward(retain_graph=True)
    (ref_w * drw).sum().backward()
    dqkv = torch.zeros_like(qkv)
    dq = dqkv[:, :, :c]
    gh, gw = ops_tfm.relpos_bwd(q, dq, heads, sh, sw, th, tw, drh, drw, True)
    tol = 1e-4 if dtype == torch.float32 else 1e-2        # bf16: dq is rounded on store
    assert rel_err(dq, ql.grad) < tol
    # bf16 at 64 x 64 and for windows up to 16 x 16: the table gradients come off the matrix cores with the logit gradients rounded to bf16 (as the reference's
    # autocast einsum backward does); everything else is fp32 FMA
    ttol = 5e-3 if (dtype == torch.bfloat16 and (grid == (64, 64) or max(grid) <= 16)) else 1e-4
    assert rel_err(gh, th.grad) < ttol and rel_err(gw, tw.grad) < ttol
    assert float(dqkv[:, :, c:].abs().sum()) == 0.0       # only the q slice is touched


def test_sam_encoder_arena_direct_gradients_equal_autograd_gradients():
    """With the flat gradient arena attached, linear / LayerNorm / relative-position-table gradients are written in
    place by the kernels; they must equal the plain autograd path (and the rel-pos tables go through the privatised
    atomic copies either way)."""
    from simpleaicv_pytorch_training_examples_amd import engine
    fx = load_golden('sam_encoder_tiny')
    a = _sam_model(fx)
    b = _sam_model(fx)
    arena = engine.FlatArena(list(b.named_parameters()), torch.device('cuda'))
    x, probe = _sam_inputs(fx)
    (a(x) * probe).sum().backward()
    arena.zero_grad()
    (b(x) * probe).sum().backward()
    torch.cuda.synchronize()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert rel_err(pb.grad, pa.grad) < 1e-4, n
