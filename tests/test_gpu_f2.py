"""SURVEY.md 8f rank 2 -- backbones that reuse the hot-path blocks -- against fixtures the reference produced
(oracle/make_golden_f2.py): the multi-scale detection ResNet-50 backbone (reference
detection/models/backbones/resnet.py:27) and the MAE pre-training model (masked_image_modeling/models/vit_mae.py:371,
whose decoder runs attention at head dim 32).  fp32 parity mode, north_star tolerance 1e-3 on outputs / loss."""
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _check_grads(model, fx, tol_norm, tol_sample):
    worst = 0.0
    for n, p in model.named_parameters():
        if n not in fx['grad_norm']:
            assert p.grad is None or not p.requires_grad, n
            continue
        assert p.grad is not None, n
        ref_n = fx['grad_norm'][n]
        assert abs(float(p.grad.norm()) - ref_n) <= tol_norm * max(ref_n, 1e-6), (n, float(p.grad.norm()), ref_n)
        if ref_n > 1e-7:
            e = rel_err(p.grad.flatten()[:64], fx['grad_sample'][n])
            worst = max(worst, e)
            assert e < tol_sample, (n, e)
    return worst


def test_detection_resnet50_backbone_matches_reference():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.backbones import resnet50backbone
    fx = load_golden('det_resnet50backbone')
    torch.manual_seed(fx['model_seed'])
    m = resnet50backbone().cuda().train()
    assert m.out_channels == fx['out_channels']
    g = torch.Generator().manual_seed(fx['data_seed'])
    x = torch.randn(*fx['shape'], generator=g).permute(0, 3, 1, 2)
    assert abs(float(x.double().sum()) - fx['input_checksum']) < 1e-6
    probes = [torch.randn(o.shape, generator=g).cuda() for o in fx['outputs']]
    outs = m(x.cuda())
    assert len(outs) == 4
    sum((o.float() * p).sum() for o, p in zip(outs, probes)).backward()
    torch.cuda.synchronize()
    for o, r in zip(outs, fx['outputs']):
        assert o.shape == r.shape and rel_err(o, r) < 1e-3
    # batch 2: BatchNorm backward + ReLU sign flips; the gate follows the reference's own reorder noise (generator)
    noise = fx['reference_noise']['fp32_reorder_grad_sample']      # 0.14: C5 is 2 x 2, BatchNorm over 8 samples
    worst = _check_grads(m, fx, max(1e-2, 0.5 * noise), max(5e-2, 2 * noise))
    for n, b in m.named_buffers():
        if n in fx['buffers_after'] and b.dtype.is_floating_point:
            assert rel_err(b, fx['buffers_after'][n]) < 1e-3, n
    print(f'det_resnet50backbone fp32: worst gradient-sample error {worst:.2e}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_mae_pretrain_model_matches_reference(dtype):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.masked_image_modeling.losses import MSELoss
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.masked_image_modeling.models.vit_mae import VITMAEPretrainModel
    fx = load_golden('mae_tiny')
    torch.manual_seed(fx['model_seed'])
    m = VITMAEPretrainModel(**fx['kwargs']).cuda().train()
    g = torch.Generator().manual_seed(fx['data_seed'])
    x = torch.randn(fx['batch'], 3, 64, 64, generator=g)
    assert abs(float(x.double().sum()) - fx['input_checksum']) < 1e-6
    torch.manual_seed(fx['noise_seed'])
    noise = torch.rand(fx['batch'], (64 // 16) ** 2)            # the reference's CPU draws, replayed
    x = x.cuda()
    if dtype == torch.bfloat16:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            pred, mask = m(x, noise)
    else:
        pred, mask = m(x, noise)
    loss = MSELoss()(pred, m.images_to_patch(x), mask)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(mask.cpu(), fx['mask'])                   # same patches kept / removed
    if dtype == torch.float32:
        assert rel_err(pred, fx['pred']) < 1e-3
        assert abs(float(loss) - fx['loss']) < 1e-3 * abs(fx['loss'])
        worst = _check_grads(m, fx, 1e-2, 2e-2)
        print(f'mae_tiny fp32: worst gradient-sample error {worst:.2e}')
    else:
        assert rel_err(pred.float(), fx['pred']) < 5e-2
        assert abs(float(loss) - fx['loss']) < 2e-2 * abs(fx['loss'])
        for n, p in m.named_parameters():
            if p.requires_grad:
                assert p.grad is not None and torch.isfinite(p.grad).all(), n


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_detection_vit_backbone_and_pyramid_neck_match_reference(dtype):
    """ViTBackbone + VitPyramidNeck (reference detection/models/backbones/vit.py:27,118): the trunk on the fused pre-LN
    blocks without class token, the neck's 2x2 stride-2 transposed convolutions as GEMMs + depth-to-space.  Same seed
    -> same initial weights as the reference (checksums in the fixture)."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.backbones.vit import ViTBackbone, VitPyramidNeck
    fx = load_golden('det_vitbackbone_tiny')
    torch.manual_seed(fx['model_seed'])
    m = ViTBackbone(**fx['kwargs']).train()
    neck = VitPyramidNeck(*fx['neck']).train()
    for prefix, mod in (('backbone.', m), ('neck.', neck)):
        for n, p in mod.named_parameters():
            s, a = fx['init_checksums'][prefix + n]
            assert abs(float(p.double().sum()) - s) < 1e-9 * max(1.0, a) and abs(float(p.double().abs().sum()) - a) < 1e-9 * max(1.0, a), n
    m, neck = m.cuda(), neck.cuda()
    g = torch.Generator().manual_seed(fx['data_seed'])
    x = torch.randn(*fx['shape'], generator=g).permute(0, 3, 1, 2)
    assert abs(float(x.double().sum()) - fx['input_checksum']) < 1e-6
    probes = [torch.randn(o.shape, generator=g).cuda() for o in fx['outputs']]
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        feat = m(x.cuda())
        outs = neck(feat)
    sum((o.float() * p).sum() for o, p in zip(outs, probes)).backward()
    torch.cuda.synchronize()
    tol = 1e-3 if dtype == torch.float32 else 3e-2
    assert feat.shape == fx['feature'].shape and rel_err(feat.float(), fx['feature']) < tol
    for o, r in zip(outs, fx['outputs']):
        assert o.shape == r.shape and rel_err(o.float(), r) < tol
    worst = 0.0
    for prefix, mod in (('backbone.', m), ('neck.', neck)):
        for n, p in mod.named_parameters():
            assert p.grad is not None, n
            ref_n = fx['grad_norm'][prefix + n]
            assert abs(float(p.grad.norm()) - ref_n) <= (5e-3 if dtype == torch.float32 else 5e-2) * max(ref_n, 1e-6), (n, float(p.grad.norm()), ref_n)
            if ref_n > 1e-7:
                e = rel_err(p.grad.flatten()[:64], fx['grad_sample'][prefix + n])
                worst = max(worst, e)
                assert e < (5e-3 if dtype == torch.float32 else 8e-2), (n, e)
    print(f'det_vitbackbone_tiny {dtype}: worst gradient-sample error {worst:.2e}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('case', ['mlp_train', 'swiglu_eval'])
def test_dinov3_backbone_matches_reference(case, dtype):
    """DinoVisionTransformer (reference detection/models/backbones/dinov3vit.py:453-571: RoPE, LayerScale, GELU-MLP / SwiGLU, the
    k-bias mask) against reference-generated outputs and gradients (oracle/make_golden_r04.py dinov3_tiny).  The weights are
    rebuilt from the same seeds: construction draw order, the periods buffer and the bias mask are checked through per-parameter
    checksums first.  'mlp_train' runs in training mode: the RoPE rescale augmentation must consume the same host draw."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.backbones.dinov3vit import DinoVisionTransformer
    fx = load_golden('dinov3_tiny')['cases'][case]
    torch.manual_seed(0)
    m = DinoVisionTransformer(**fx['kwargs'])
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('.gamma'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3 + 1.0)
            elif n.endswith('.bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    sd = m.state_dict()
    assert list(sd.keys()) == list(fx['param_sum'].keys())
    for k, v in sd.items():
        assert abs(float(v.double().sum()) - fx['param_sum'][k]) <= 1e-9 * max(1.0, fx['param_abs_sum'][k]), k
    x = torch.randn(2, 3, fx['hw'][0], fx['hw'][1], generator=g)
    assert abs(float(x.double().sum()) - fx['input_checksum']) < 1e-6
    m = m.cuda().train(fx['train'])
    xg = x.cuda()
    torch.manual_seed(5)
    if dtype == torch.bfloat16:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = m(xg)
    else:
        out = m(xg)
    probe = torch.randn(fx['out'].shape, generator=g)
    (out.float() * probe.cuda()).sum().backward()
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    assert out.shape == fx['out'].shape and out.is_contiguous()
    assert rel_err(out.float(), fx['out']) < (1e-3 if f32 else 4e-2)
    # (no image gradient: the patch embedding is the first node of a training graph and produces none, as in the other ViT trunks)
    worst = 0.0
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        ref = fx['grad_sample'][n]
        got = p.grad.flatten()[:64].float().cpu()
        scale = max(float(ref.abs().max()), 1e-3 * fx['grad_norm'][n], 1e-12)
        err = float((got - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err < (2e-3 if f32 else 1.5e-1), (n, err)
        assert abs(float(p.grad.float().norm()) - fx['grad_norm'][n]) <= (2e-3 if f32 else 8e-2) * max(fx['grad_norm'][n], 1e-9), n
    k3 = m.blocks[0].attn.qkv.bias.grad[128:256]
    assert float(k3.abs().max()) == 0.0                            # the k third of the qkv bias is masked: no gradient
    print(f'dinov3_tiny {case} {"fp32" if f32 else "bf16"}: worst gradient-sample error {worst:.2e}')


def test_dinov3_factories_and_refusals():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import backbones
    m = backbones.__dict__['dinov3_vit_small_plus_patch16_backbone']()
    assert m.out_channels == 384 and len(m.blocks) == 12 and m.blocks[0].mlp.w1.out_features == 1536
    assert sum(p.numel() for p in m.parameters()) == sum(p.numel() for p in backbones.dinov3_vit_small_plus_patch16_backbone().parameters())
    with pytest.raises(NotImplementedError, match='head dim 128'):
        backbones.dinov3vit.SelfAttention(4096, head_nums=32)          # the 7B model's geometry


def _dinov3_detector(case):
    """RetinaNet / FCOS on the two-block DINOv3 trunk of the fixture (oracle/make_golden_r04.py dinov3_detectors), rebuilt from its seeds"""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import backbones, dinov3_vit_fcos, dinov3_vit_retinanet
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.backbones.dinov3vit import DinoVisionTransformer
    gold = load_golden('dinov3_detectors')
    backbones.__dict__['tiny_dinov3_backbone'] = lambda pretrained_path='', **kw: DinoVisionTransformer(**gold['trunk'], **kw)
    torch.manual_seed(0)
    m = (dinov3_vit_retinanet.RetinaNet if case == 'retinanet' else dinov3_vit_fcos.FCOS)('tiny_dinov3_backbone', planes=64, num_classes=6)
    g = torch.Generator().manual_seed(44)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('.gamma'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3 + 1.0)
            elif n.endswith('.bias') and 'cls_out' not in n and 'cls_head' not in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    x = torch.randn(2, 3, 128, 96, generator=g)
    return gold['cases'][case], m, x, g


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('case', ['retinanet', 'fcos'])
def test_dinov3_vit_detectors_match_reference(case, dtype):
    """dinov3_vit_retinanet.RetinaNet / dinov3_vit_fcos.FCOS (reference detection/models/dinov3_vit_retinanet.py:28-112,
    dinov3_vit_fcos.py:28-101): DINOv3 trunk -> VitPyramidNeck -> RetinaFPN -> shared towers.  Construction (keys + every initial
    tensor by checksum), every level's outputs and every parameter gradient against what the reference produced; training mode, so
    the trunk's RoPE rescale draw is replayed."""
    fx, m, x, g = _dinov3_detector(case)
    sd = m.state_dict()
    assert list(sd.keys()) == list(fx['param_sum'].keys())
    for k, v in sd.items():
        assert abs(float(v.double().sum()) - fx['param_sum'][k]) <= 1e-6 * max(1.0, fx['param_abs_sum'][k]), k
    assert abs(float(x.double().sum()) - fx['input_checksum']) < 1e-6
    m = m.cuda().train()
    f32 = dtype == torch.float32
    torch.manual_seed(5)
    if f32:
        outs = m(x.cuda())
    else:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            outs = m(x.cuda())
    assert [len(group) for group in outs] == fx['groups']
    flat = [o for group in outs for o in group]
    probes = [torch.randn(o.shape, generator=g) for o in fx['outs']]
    for o, ref in zip(flat, fx['outs']):
        assert tuple(o.shape) == tuple(ref.shape)
        assert rel_err(o.float().cpu(), ref) < (1e-3 if f32 else 5e-2)
    sum((o.float() * p.cuda()).sum() for o, p in zip(flat, probes)).backward()
    torch.cuda.synchronize()
    worst, far = 0.0, 0
    params = dict(m.named_parameters())
    assert sorted(n for n, p in params.items() if p.grad is None) == sorted(fx['no_grad'])      # the stride-4 neck branch is unused
    for n, ref_n in fx['grad_norm'].items():
        p = params[n]
        gn = float(p.grad.float().norm())
        # bf16, FCOS: the extra pyramid levels P6 / P7 are 2 x 2 and 1 x 1 maps at this input size and the head's GroupNorm(32, 64)
        # normalises groups of 8 and 2 VALUES there -- with two values x_hat is +-1 whatever they are, the gradient through it is pure
        # rounding (it is exactly 0 in exact arithmetic up to eps), and everything upstream of those levels only (the P6 / P7 convolutions)
        # inherits it: their norms are pinned by the fp32 case (2e-2), in bf16 only their samples are (below)
        rounding_only = (not f32) and case == 'fcos' and n.startswith(('fpn.P6', 'fpn.P7'))
        if not rounding_only:
            assert abs(gn - ref_n) <= (2e-2 if f32 else 1.5e-1) * max(ref_n, 1e-6) + 1e-7, (n, gn, ref_n)
        ref = fx['grad_sample'][n]
        got = p.grad.flatten()[:64].float().cpu()
        scale = max(float(ref.abs().max()), 1e-2 * ref_n, 1e-12)
        err = float((got - ref).abs().max()) / scale
        worst = max(worst, err)
        # FCOS: GroupNorm(32, 64) on the 2 x 2 and 1 x 1 levels normalises groups of 8 and 2 values -- rounding is amplified there, so
        # bf16 may leave a few tensors' samples beyond 0.3 of the gradient scale (none beyond the scale); fp32 stays within 2e-2
        assert err <= (2e-2 if f32 else 1.0), (n, err)
        far += err > 3e-1
    assert far <= 0.05 * len(fx['grad_norm']), far
    print(f'dinov3 {case} {"fp32" if f32 else "bf16"}: worst gradient-sample error {worst:.2e}')
