"""Round-4 GPU parity tests.

1. float64 arbiter for whole-model gradients.  tests/golden/fp64_<name>.pt (oracle/make_golden_fp64.py)
   holds, per parameter, the gradient of the REFERENCE run in float64 at up to 1024 evenly spaced positions and how far the
   reference's own fp32 run is from it (relative L2 over those positions).  The HIP fp32 path is held to the same yardstick:
       err_hip[n] = || hip[idx] - g64 || / || g64 ||
   through the three gates of _arbiter() below (aggregate and median at 1.5 x the reference's fp32 run, every single parameter at
   3 x the worse of the reference's own two fp32 runs; no "k of n may fail" allowance anywhere).
   Two regimes, both visible in the fixtures:
     * BatchNorm networks at initialisation (ResNet-50, DETR-R50): fp32 itself is 1-2 % away from float64 (rounding flips ReLU
       signs, batch statistics amplify it); K_REF x that is the gate, the floor is irrelevant.
     * LayerNorm transformers (ViT-B, SAM encoder): the reference's fp32 run is 3e-7..2e-6 from float64 -- it evaluates erf / exp
       through libm, sums in one fixed order.  The HIP path is exact-product fp32 MFMA with a different summation tree, a
       one-exp / one-rcp erf (|error| <= 1.5e-7 absolute, csrc/common.h) and v_exp_f32 softmax: its distance to float64 is set by
       those, not by the reference's, so the floor carries the gate there.  FLOOR = 1e-4 is 10 x below the 1e-3 parity bound of
       north_star and ~50 x above what the fp32 MFMA path shows (printed by each test).
2. ConvBnActBlock(has_bn=False) and the depthwise form against reference-generated outputs / gradients
   (oracle/make_golden_r04.py: convbnact_variants) -- VERDICT r03 item 8.
3. SAM Block whose relative-position tables were built for another grid (get_rel_pos interpolation, image_encoder.py:96-103)
   against the reference Block (sam_block_relpos_resized) -- VERDICT r03 items 6/8.
"""
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

K_REF = 1.5        # |HIP - fp64| <= 1.5 x |ref_fp32 - fp64| -- held for the aggregate and for the per-parameter median
K_PARAM = 3.0      # every single parameter: <= 3 x the worse of the reference's own two fp32 runs for THAT parameter (see below)
FLOOR = 1e-4       # absolute floor of the relative-L2 gate (see the module docstring)


def _arbiter(model_grads, fp64, label):
    """model_grads: {name: fp32 gradient tensor on the GPU}.  Three gates, none with a "k of n may fail" allowance:
      (1) all sampled gradient elements of all parameters together:  err_hip <= K_REF * err_ref + FLOOR;
      (2) the median over the parameters of err_hip[n] / err_ref[n]  <= K_REF;
      (3) EVERY parameter: err_hip[n] <= K_PARAM * max(err_ref[n], err_ref_alt[n]) + FLOOR.
    Why (3) is not 1.5 x a single reference run: the ratio of two rounding-noise magnitudes is heavy-tailed per tensor.  The
    reference's OWN second fp32 run (channels_last / 3 threads; stored as ref32_alt_l2) breaks a 1.5 x gate against its first
    run for 5 of 161 ResNet-50 parameters (max 1.72 x) and for 343 of 349 DETR parameters (median 3.3 x: its 3-thread run is
    simply farther from float64) -- a gate the reference fails against itself cannot certify anything.  Measured for the HIP
    path (r04c, one box): ResNet-50 b32 aggregate 1.13 x (median 1.13, max 2.72), ViT-B 0.48 x, SAM-B encoder 1.97 x of
    7e-7, DETR-R50 0.54 x (median 0.53, max 3.96 vs the first run alone).  A systematic error of 1e-5 in any tensor of the two
    LayerNorm models would fail (1) and (3) there; on the BatchNorm models fp32 itself is 1-2 % from float64, so only the
    per-kernel tests (test_gpu_kernels.py, 1e-4) can see anything smaller -- no whole-model gate can, with any arbiter."""
    bad, ratios = [], []
    cat_h, cat_r = [], []
    for n, g64 in fp64['g64'].items():
        assert n in model_grads, n
        h = model_grads[n].detach().flatten().cpu()[fp64['idx'][n]].double()
        den = float(g64.norm())
        if den < 1e-12:
            continue
        err = float((h - g64).norm()) / den
        ref = max(fp64['ref32_l2'][n], fp64['ref32_alt_l2'].get(n, 0.0))
        gate = K_PARAM * ref + FLOOR
        ratios.append(err / max(fp64['ref32_l2'][n], 1e-12))
        cat_h.append(h)
        cat_r.append(g64)
        if err > gate:
            bad.append((n, err, gate))
    cat_h, cat_r = torch.cat(cat_h), torch.cat(cat_r)
    all_err = float((cat_h - cat_r).norm() / cat_r.norm())
    ratios.sort()
    median = ratios[len(ratios) // 2]
    print(f'{label}: HIP fp32 vs float64 over all sampled gradient elements {all_err:.3e} (reference fp32 vs float64 '
          f'{fp64["all_l2"]["ref32"]:.3e}); per-parameter err / reference err: median {median:.2f} '
          f'max {ratios[-1]:.2f}; {sum(r > K_REF for r in ratios)} of {len(ratios)} above {K_REF} x the first reference run, '
          f'{len(bad)} over the per-parameter gate')
    assert all_err <= K_REF * fp64['all_l2']['ref32'] + FLOOR, (all_err, fp64['all_l2'])
    assert median <= K_REF or all_err <= FLOOR, median
    assert not bad, bad[:8]


def test_resnet50_b32_gradients_against_the_float64_arbiter():
    from test_gpu_models import _build
    fx, f64 = load_golden('resnet50_b32_112'), load_golden('fp64_resnet50_b32_112')
    model, crit, x, y = _build(fx, 'resnet50')
    model.train()
    logits = model(x)
    loss = crit(logits, y)
    loss.backward()
    torch.cuda.synchronize()
    lg_err = float((logits.double().cpu() - f64['logits64']).norm() / f64['logits64'].norm())
    print(f'resnet50_b32_112: logits vs float64 {lg_err:.3e} (reference fp32: {f64["ref32_logits_l2"]:.3e})')
    assert lg_err <= K_REF * f64['ref32_logits_l2'] + FLOOR
    assert abs(float(loss) - f64['loss64']) <= (K_REF * f64['ref32_loss_rel'] + FLOOR) * abs(f64['loss64'])
    _arbiter({n: p.grad for n, p in model.named_parameters()}, f64, 'resnet50_b32_112')


def test_vit_base_gradients_against_the_float64_arbiter():
    from test_gpu_models import _build_vit
    fx, f64 = load_golden('vit_base_patch16_b2_224'), load_golden('fp64_vit_base_patch16_b2_224')
    model, crit, x, y = _build_vit(fx, 'vit_base_patch16')
    model.train()
    logits = model(x)
    crit(logits, y).backward()
    torch.cuda.synchronize()
    _arbiter({n: p.grad for n, p in model.named_parameters()}, f64, 'vit_base_patch16_b2_224')


def test_sam_b_encoder_gradients_against_the_float64_arbiter():
    from test_gpu_sam import _sam_inputs, _sam_model
    fx, f64 = load_golden('sam_b_encoder_256'), load_golden('fp64_sam_b_encoder_256')
    m = _sam_model(fx)
    x, probe = _sam_inputs(fx)
    out = m(x)
    (out * probe).sum().backward()
    torch.cuda.synchronize()
    sub = out.detach().double().cpu()[:, ::8, ::4, ::4]
    o_err = float((sub - f64['output64_sub']).norm() / f64['output64_sub'].norm())
    print(f'sam_b_encoder_256: output vs float64 {o_err:.3e} (reference fp32: {f64["ref32_output_l2"]:.3e})')
    assert o_err <= K_REF * f64['ref32_output_l2'] + FLOOR
    _arbiter({n: p.grad for n, p in m.named_parameters()}, f64, 'sam_b_encoder_256')


def test_detr_r50_gradients_against_the_float64_arbiter():
    from test_gpu_detr import _build
    fx, f64 = load_golden('detr_r50_small'), load_golden('fp64_detr_r50_small')
    m, crit, images, masks, annots = _build(fx)
    cls_out, reg_out = m(images, masks)
    total = sum(crit([cls_out, reg_out], annots).values())
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total) - f64['total64']) <= (K_REF * f64['ref32_total_rel'] + FLOOR) * abs(f64['total64'])
    _arbiter({n: p.grad for n, p in m.named_parameters()}, f64, 'detr_r50_small')


# ------------------------------------------------------------------------------------------------ ConvBnActBlock variants
def _l2_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize('case', ['bias_relu_3x3', 'bias_only_1x1_s2', 'depthwise_bn_relu'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_convbnact_block_variants_match_reference(case, dtype):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import ConvBnActBlock
    c = load_golden('convbnact_variants')['cases'][case]
    blk = ConvBnActBlock(**c['kwargs'])
    assert list(blk.state_dict().keys()) == list(c['state_dict'].keys())          # the drop-in contract: same keys, same order
    blk.load_state_dict(c['state_dict'])
    blk = blk.cuda().train()
    x = c['x'].cuda().requires_grad_(True)
    if dtype == torch.float32:
        out = blk(x)
    else:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = blk(x)
    (out.float() * c['probe'].cuda()).sum().backward()
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    # fp32: max-abs error on the tensor's scale (north_star 1e-3).  bf16: relative L2 -- a pre-activation that rounds across zero
    # flips one ReLU gate and moves single gradient elements by O(1), which a max-abs metric reads as a 6-26 % error of a tensor
    # that is right to 1-2 % everywhere else
    err = rel_err if f32 else _l2_err
    assert err(out.float(), c['out']) < (1e-3 if f32 else 2e-2)
    assert err(x.grad, c['dx']) < (1e-3 if f32 else 6e-2)
    for n, p in blk.named_parameters():
        assert p.grad is not None, n
        assert err(p.grad, c['grads'][n]) < (1e-3 if f32 else 6e-2), n
    if f32:
        for n, b in blk.named_buffers():
            if b.dtype.is_floating_point:
                assert rel_err(b, c['buffers_after'][n]) < 1e-3, n
            else:
                assert int(b) == int(c['buffers_after'][n]), n


def test_convbnact_block_still_refuses_general_grouped_convolutions():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import ConvBnActBlock
    with pytest.raises(NotImplementedError, match='groups=4'):
        ConvBnActBlock(32, 32, 3, 1, 1, groups=4)


# ------------------------------------------------------------------------------------------------ resized rel-pos tables
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_sam_block_with_interpolated_relative_position_tables(dtype):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import Block
    fx = load_golden('sam_block_relpos_resized')
    blk = Block(inplanes=128, head_nums=2, mlp_ratio=4.0, input_size=(8, 8), window_size=0)
    blk.load_state_dict(fx['state_dict'])
    blk = blk.cuda().train()
    x = fx['x'].cuda().requires_grad_(True)
    if dtype == torch.float32:
        out = blk(x)
    else:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = blk(x)
    (out.float() * fx['probe'].cuda()).sum().backward()
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    assert rel_err(out.float(), fx['out']) < (1e-3 if f32 else 3e-2)
    assert rel_err(x.grad, fx['dx']) < (1e-3 if f32 else 5e-2)
    for n, p in blk.named_parameters():
        assert p.grad is not None, n
        assert p.grad.shape == fx['grads'][n].shape, n           # the tables keep their 15 rows; the gradient comes back through A^T
        assert rel_err(p.grad, fx['grads'][n]) < (2e-3 if f32 else 8e-2), n


# ------------------------------------------------------------------------------------------------ fused stem: BN + ReLU + MaxPool
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(4, 3, 64, 64), (2, 3, 50, 70), (3, 3, 33, 47)])
def test_stem_bn_relu_maxpool_fused_equals_the_unfused_pair_and_torch(shape, dtype, monkeypatch):
    """ResNet stem: conv 7x7/2 -> BatchNorm -> ReLU -> MaxPool(3, 2, 1) (reference resnet.py:172-184, 226-229) with the last three
    as ONE pass over the convolution output (csrc/pool.hip bn_relu_maxpool_*): output, running statistics and every gradient
    against (a) the same block with the unfused kernels (SAICV_STEM_POOL_FUSE=0) and (b) torch's own modules on the CPU in fp32.
    Odd extents exercise the border windows and the pixels no window covers."""
    import torch.nn as nn
    from simpleaicv_pytorch_training_examples_amd import ops
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import ConvBnActBlock
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    ref = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
    with torch.no_grad():
        ref[1].weight.copy_(torch.rand(64, generator=g) + 0.5)
        ref[1].bias.copy_(torch.randn(64, generator=g) * 0.3)
    out_ref = ref(x)
    probe = torch.randn(out_ref.shape, generator=g)
    (out_ref * probe).sum().backward()

    def run(fuse):
        monkeypatch.setattr(ops, 'STEM_POOL_FUSE', fuse)
        blk = ConvBnActBlock(3, 64, 7, 2, 3)
        blk.layer[0].weight.data.copy_(ref[0].weight.data)
        blk.layer[1].load_state_dict({k: v for k, v in ref[1].state_dict().items()})
        blk.layer[1].running_mean.zero_()
        blk.layer[1].running_var.fill_(1.0)
        blk.layer[1].num_batches_tracked.zero_()
        blk = blk.cuda().train()
        ctx = torch.autocast('cuda', dtype=torch.bfloat16) if dtype == torch.bfloat16 else torch.autocast('cuda', enabled=False)
        with ctx:
            xin = ops.pack_stem_input(x.cuda(), blk.layer[0])
            if fuse:
                out = blk(xin, pool=(3, 2, 1))
            else:
                out = ops.max_pool2d(blk(xin), 3, 2, 1)
        (out.float() * probe.cuda()).sum().backward()
        torch.cuda.synchronize()
        return out.float(), {n: p.grad.clone() for n, p in blk.named_parameters()}, blk.layer[1].running_var.clone()

    out_f, g_f, rv_f = run(True)
    out_u, g_u, rv_u = run(False)
    f32 = dtype == torch.float32
    # fused vs unfused: the same arithmetic on the same rounded values -> the pooled maxima agree to the last bit in fp32 (the
    # affine is one FMA in both) and up to rare one-ulp bf16 ties otherwise; gradients differ only by summation order
    assert rel_err(out_f, out_u) < (1e-6 if f32 else 8e-3)
    assert rel_err(rv_f, rv_u) < 1e-6
    for n in g_f:
        assert _l2_err(g_f[n], g_u[n]) < (2e-5 if f32 else 2e-2), n
    # against torch (fp32 CPU)
    assert rel_err(out_f, out_ref) < (1e-3 if f32 else 3e-2)
    names = {'layer.0.weight': ref[0].weight.grad, 'layer.1.weight': ref[1].weight.grad, 'layer.1.bias': ref[1].bias.grad}
    for n, r in names.items():
        assert _l2_err(g_f[n], r) < (1e-3 if f32 else 1.5e-1), n        # bf16 against an fp32 reference: the conv output is rounded before the statistics
    assert rel_err(rv_f, ref[1].running_var) < (1e-3 if f32 else 1e-2)


# ------------------------------------------------------------------------------------------------ shortcut BatchNorm applied in the join
def _torch_block(kind, inplanes, planes, stride):
    """plain-torch restatement of the reference BasicBlock / Bottleneck (resnet.py:51-97, :100-155) for the CPU side"""
    import torch.nn as nn

    def cba(i, o, k, s, p, act):
        return nn.Sequential(nn.Conv2d(i, o, k, s, p, bias=False), nn.BatchNorm2d(o), nn.ReLU() if act else nn.Sequential())

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            if kind == 'basic':
                self.conv1, self.conv2 = cba(inplanes, planes, 3, stride, 1, True), cba(planes, planes, 3, 1, 1, False)
                out = planes
            else:
                self.conv1, self.conv2 = cba(inplanes, planes, 1, 1, 0, True), cba(planes, planes, 3, stride, 1, True)
                self.conv3 = cba(planes, planes * 4, 1, 1, 0, False)
                out = planes * 4
            self.downsample_conv = cba(inplanes, out, 1, stride, 0, False)

        def forward(self, x):
            y = self.conv2(self.conv1(x))
            if kind != 'basic':
                y = self.conv3(y)
            return torch.relu(y + self.downsample_conv(x))
    return Blk()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('kind,inplanes,planes,stride,hw', [('bottleneck', 64, 64, 1, 14), ('bottleneck', 256, 128, 2, 14),
                                                             ('basic', 64, 128, 2, 12), ('bottleneck', 1024, 768, 2, 8)])
@pytest.mark.parametrize('inline', [True, False], ids=['atomic_rows', 'finalize'])
def test_shortcut_batchnorm_applied_inside_the_join(kind, inplanes, planes, stride, hw, dtype, inline, monkeypatch):
    """`identity = downsample_conv(x); x = relu(x + identity)` (reference resnet.py:90-95, :148-153) with the shortcut's
    BatchNorm-apply moved into the join pass (ops.DS_JOIN_FUSE, csrc/bn.hip bn_act_fwd_join): output, running statistics of BOTH
    BatchNorms and every gradient against (a) the materialised form (SAICV_DS_JOIN_FUSE=0) and (b) torch modules on the CPU.
    The last case (3072 output channels) takes the finalize-kernel form of the main branch even with atomic rows on."""
    from simpleaicv_pytorch_training_examples_amd import ops
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import BasicBlock, Bottleneck
    monkeypatch.setattr(ops, 'BN_INLINE', inline)
    g = torch.Generator().manual_seed(inplanes + planes + hw)
    ref = _torch_block(kind, inplanes, planes, stride)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.3 + (1.0 if n.endswith('1.weight') else 0.0))
    x = torch.randn(4, inplanes, hw, hw, generator=g).to(dtype).float()          # both sides see the same (rounded) input
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    out_ref = ref(xr)
    probe = torch.randn(out_ref.shape, generator=g)
    (out_ref * probe).sum().backward()
    ref_grads = {n.replace('.0.', '.layer.0.').replace('.1.', '.layer.1.'): p.grad for n, p in ref.named_parameters()}
    ref_bufs = {n.replace('.1.', '.layer.1.'): b for n, b in ref.named_buffers()}

    def run(fuse):
        monkeypatch.setattr(ops, 'DS_JOIN_FUSE', fuse)
        blk = (BasicBlock if kind == 'basic' else Bottleneck)(inplanes, planes, stride)
        blk.load_state_dict({k.replace('.0.', '.layer.0.').replace('.1.', '.layer.1.'): v for k, v in sd0.items()})
        for m in blk.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        blk = blk.cuda().train()
        xin = x.cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ctx = torch.autocast('cuda', dtype=torch.bfloat16) if dtype == torch.bfloat16 else torch.autocast('cuda', enabled=False)
        with ctx:
            out = blk(xin)
        (out.float() * probe.cuda()).sum().backward()
        with torch.no_grad(), ctx:
            out_eval = blk.eval()(xin.detach()).float()
        torch.cuda.synchronize()
        return (out.float(), xin.grad.float(), {n: p.grad.clone() for n, p in blk.named_parameters()},
                {n: b.clone().float() for n, b in blk.named_buffers()}, out_eval)

    out_f, dx_f, g_f, b_f, ev_f = run(True)
    out_u, dx_u, g_u, b_u, ev_u = run(False)
    f32 = dtype == torch.float32
    # (a) fused vs materialised: in fp32 the only difference is one rounding of the normalised shortcut that no longer happens
    # (it stays in registers); in bf16 that rounding was a bf16 one, so the fused form is the MORE accurate of the two
    assert _l2_err(out_f, out_u) < (1e-6 if f32 else 4e-3)
    assert _l2_err(ev_f, ev_u) < (1e-6 if f32 else 4e-3)
    # (bf16 gradients: the dropped rounding moves a few ReLU gates of the join, 2-4 % in L2 -- both forms are equally far from fp32)
    # fp32: 2e-5 is what summation order gives; ONE join gate whose pre-activation sits within an ulp of zero flipping between
    # the two forms moves a gradient by ~1e-3 (measured on the smoke model, __graft_entry__.py) -- hence 5e-3, the outputs above
    # (which a flip at zero does not move) carry the tight bound
    assert _l2_err(dx_f, dx_u) < (5e-3 if f32 else 6e-2)
    for n in g_f:
        assert _l2_err(g_f[n], g_u[n]) < (5e-3 if f32 else 6e-2), n
    for n in b_f:
        assert _l2_err(b_f[n], b_u[n]) < (1e-5 if f32 else 1e-4), n       # (means near zero summed by atomics in a free order)
    # (b) against torch in fp32 on the CPU
    assert _l2_err(out_f, out_ref) < (1e-4 if f32 else 1e-2)
    # bf16: every activation is rounded to 8 bits before the next statistics / ReLU gate; gradients at 14 x 14 x 4 samples sit
    # 5-10 % from an fp32 run (the materialised form is as far: checked above against it at 2e-2)
    # fp32 against torch: different summation trees in the convolutions put a handful of the 1e5 ReLU gates on the other side of
    # zero; with 4 x 4 x 4 .. 14 x 14 x 4 samples per BatchNorm channel each flip is visible at the 1e-3 .. 1e-2 level
    assert _l2_err(dx_f, xr.grad) < (2e-2 if f32 else 1.5e-1)
    for n, r in ref_grads.items():
        assert _l2_err(g_f[n], r) < (2e-2 if f32 else 1.5e-1), n
    for n, r in ref_bufs.items():
        assert _l2_err(b_f[n], r.float()) < (1e-4 if f32 else 1e-2), n
