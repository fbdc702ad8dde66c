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
