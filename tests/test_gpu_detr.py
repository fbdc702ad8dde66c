"""DETR on a real MI355X against the fixture produced by the reference's own model + loss
(tests/golden/detr_r18_tiny.pt, oracle/make_golden_detr.py).  Dropout is zeroed on both sides.

fp32: class logits / boxes of all six decoder layers within 1e-3 (north_star), loss terms within 1e-3,
      Hungarian assignment identical, BN statistics within 1e-3, gradient norms within 2e-2 and samples within
      5e-2 of the gradient scale (BatchNorm at batch 4 on 6x8 feature maps amplifies rounding, cf. the ResNet
      cases of test_gpu_models.py).
bf16: measured against the reference's own bf16-vs-fp32 deviation stored in the fixture.
Dropout: the in-kernel attention dropout keeps the expectation and is reproducible from its seed.
"""
import pytest
import torch

from conftest import load_golden, rel_err
from oracle.make_golden_detr import detr_inputs, zero_dropout

pytestmark = pytest.mark.gpu


def _build(fx):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import detr
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    torch.manual_seed(fx['model_seed'])
    m = detr.__dict__[fx['factory']](**fx['kwargs'])
    zero_dropout(m)
    images, masks, annots = detr_inputs(fx['batch'], fx['data_seed'], num_classes=fx['kwargs']['num_classes'])
    assert abs(float(images.double().sum() + annots.double().sum()) - fx['input_checksum']) < 1e-6
    return m.cuda().train(), DETRLoss(num_classes=fx['kwargs']['num_classes']), images.cuda(), masks.cuda(), annots.cuda()


@pytest.mark.parametrize('name', ['detr_r18_tiny', 'detr_r50_small'])
def test_detr_fp32_matches_reference(name, deterministic):
    """detr_r50_small: resnet50_detr as the reference config builds it (real ResNet-50 backbone, 100 queries, 80 classes)
    on a 256 x 256 canvas -- fixture produced by the reference's DETR + DETRLoss (oracle/make_golden_detr.py)."""
    fx = load_golden(name)
    m, crit, images, masks, annots = _build(fx)
    cls_out, reg_out = m(images, masks)
    ld = crit([cls_out, reg_out], annots)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    assert cls_out.shape == fx['cls_outputs'].shape and reg_out.dtype == torch.float32
    assert rel_err(cls_out, fx['cls_outputs']) < 1e-3
    assert rel_err(reg_out, fx['reg_outputs']) < 1e-3
    for k, v in fx['loss'].items():
        assert abs(float(ld[k]) - v) < 1e-3 * max(abs(v), 1e-2), (k, float(ld[k]), v)
    with torch.no_grad():
        idx = crit.get_matched_pred_target_idxs(cls_out[-1].float(), torch.clamp(reg_out[-1], 1e-4, 1 - 1e-4).float(), annots)
    for (i, j), (ri, rj) in zip(idx, fx['indices']):
        assert torch.equal(i, ri) and torch.equal(j, rj)
    worst = 0.0
    # gradient samples: 5e-2, or twice what the reference moves from ITSELF under another fp32 summation order
    # (stored by the generator): batch-2 BatchNorm backward + ReLU sign flips at near-zero pre-activations
    gtol = max(5e-2, 2 * fx['reference_noise'].get('fp32_reorder_grad_sample', 0.0))
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        ref_n = fx['grad_norm'][n]
        assert abs(float(p.grad.norm()) - ref_n) <= 2e-2 * max(ref_n, 1e-6), (n, float(p.grad.norm()), ref_n)
        if ref_n > 1e-7:
            e = rel_err(p.grad.flatten()[:64], fx['grad_sample'][n])
            worst = max(worst, e)
            assert e < gtol, (n, e, gtol)
    for n, b in m.named_buffers():
        if n in fx['buffers_after'] and b.dtype.is_floating_point:
            assert rel_err(b, fx['buffers_after'][n]) < 1e-3, n
    print(f'{name} fp32: worst gradient-sample error {worst:.2e}')


@pytest.mark.parametrize('name', ['detr_r18_tiny', 'detr_r50_small'])
def test_detr_bf16_tracks_reference_autocast(name):
    fx = load_golden(name)
    m, crit, images, masks, annots = _build(fx)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        cls_out, reg_out = m(images, masks)
        ld = crit([cls_out, reg_out], annots)
    total = sum(ld.values())
    total.backward()
    torch.cuda.synchronize()
    noise = fx['reference_noise']
    assert rel_err(cls_out.float(), fx['cls_outputs']) < 1.5 * noise['bf16_cls'] + 2e-2
    assert rel_err(reg_out.float(), fx['reg_outputs']) < 1.5 * noise['bf16_reg'] + 2e-2
    assert abs(float(total) - fx['total']) < (1.5 * noise['bf16_loss'] + 2e-2) * abs(fx['total'])
    a = torch.cat([p.grad.flatten()[:64].double().cpu() for _, p in m.named_parameters()])
    b = torch.cat([fx['grad_sample'][n].double() for n, _ in m.named_parameters()])
    cos = float(a @ b / (a.norm() * b.norm()))
    assert cos > noise['bf16_grad_sample_cos'] - 0.1, cos


def test_attention_dropout_is_seeded_and_unbiased():
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    g = torch.Generator().manual_seed(0)
    b, heads, d, n = 2, 8, 32, 300
    q, k, v = (torch.randn(b, n, heads * d, generator=g).cuda() for _ in range(3))
    scale = d ** -0.5
    ref, _ = ops_tfm.sattn_fwd(q, k, v, heads, scale)
    o1, lse1 = ops_tfm.sattn_fwd(q, k, v, heads, scale, dropout_p=0.1, seed=123)
    o2, _ = ops_tfm.sattn_fwd(q, k, v, heads, scale, dropout_p=0.1, seed=123)
    o3, _ = ops_tfm.sattn_fwd(q, k, v, heads, scale, dropout_p=0.1, seed=124)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    acc = torch.zeros_like(ref)
    for s in range(64):
        acc += ops_tfm.sattn_fwd(q, k, v, heads, scale, dropout_p=0.1, seed=1000 + s)[0]
    assert rel_err(acc / 64, ref) < 0.08                       # E[dropout(P) V] = P V
    # backward regenerates the forward mask: finite-difference-free check through linearity in v
    dout = torch.randn(b, n, heads * d, generator=g).cuda()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops_tfm.sattn_bwd(q, k, v, o1, dout, lse1, heads, scale, dq, dk, dv, dropout_p=0.1, seed=123)
    v2 = torch.randn(b, n, heads * d, generator=g).cuda()
    o_v2, _ = ops_tfm.sattn_fwd(q, k, v2, heads, scale, dropout_p=0.1, seed=123)
    # out is linear in v with the mask fixed:  <dout, out(v2)> == <dv, v2>
    lhs = float((dout.double() * o_v2.double()).sum())
    rhs = float((dv.double() * v2.double()).sum())
    assert abs(lhs - rhs) < 2e-3 * max(abs(lhs), 1.0), (lhs, rhs)


def test_train_detection_loop_runs():
    import logging
    import numpy as np
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.common import DETRDetectionCollater
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import detr
    from simpleaicv_pytorch_training_examples_amd.tools import scripts, utils

    class Set(torch.utils.data.Dataset):
        def __len__(self):
            return 8

        def __getitem__(self, i):
            rng = np.random.RandomState(i)
            h, w = 160, 192
            n = 2 + i % 3
            x1y1 = rng.uniform(10, 80, (n, 2))
            wh = rng.uniform(20, 70, (n, 2))
            a = np.concatenate([x1y1, x1y1 + wh, rng.randint(0, 20, (n, 1))], axis=1).astype(np.float32)
            return {'image': rng.randn(h, w, 3).astype(np.float32), 'annots': a, 'scale': 1.0, 'size': [h, w]}

    class config:
        pass
    torch.manual_seed(0)
    config.network = 'resnet18_detr'
    config.model = detr.resnet18_detr(num_classes=20, query_nums=20)
    config.train_criterion = DETRLoss(num_classes=20)
    config.train_dataset = Set()
    config.batch_size = 4
    config.accumulation_steps = 1
    config.optimizer = ('AdamW', {'lr': 1e-4, 'global_weight_decay': False, 'weight_decay': 1e-4,
                                  'no_weight_decay_layer_name_list': []})
    config.scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [100]})
    config.epochs = 1
    config.print_interval = 1
    config.use_amp = True
    config.use_ema_model = False
    config.clip_max_norm = 0.1
    config.local_rank = 0
    config.group = None
    config.gpus_num = 1
    config.sync_bn = False
    model = config.model.cuda()
    optimizer, _ = utils.build_optimizer(config, model)
    scheduler = utils.Scheduler(config, optimizer)
    model, config.ema_model, config.scaler = utils.build_training_mode(config, model)
    before = model.arena.flat_param.clone()
    loader = torch.utils.data.DataLoader(config.train_dataset, batch_size=4, shuffle=False, drop_last=True,
                                         collate_fn=DETRDetectionCollater(resize=192, resize_type='yolo_style'))
    logger = logging.getLogger('saicv_test_detr')
    logger.setLevel(logging.INFO)
    records = []
    handler = logging.Handler()
    handler.emit = lambda r: records.append(r.getMessage())
    logger.addHandler(handler)
    loss = scripts.train_detection(loader, model, config.train_criterion, optimizer, scheduler, 1, logger, config)
    text = '\n'.join(records)
    assert loss > 0 and loss == loss
    assert 'train: epoch 0001, iter [00002, 00002]' in text and 'total_loss:' in text and 'layer_5_box_iou_loss:' in text
    assert not torch.equal(before, model.arena.flat_param) and torch.isfinite(model.arena.flat_param).all()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_detr_backbone_stem_and_layer1_at_800x1333_match_reference(dtype):
    """BASELINE.json configs[3] at its own resolution: conv1 (the space-to-depth stem) + maxpool1 + layer1 of
    detr_resnet50backbone on one 3 x 800 x 1333 image (odd width: 667 / 334 columns after the two stride-2 stages), forward
    and backward -- fixture produced by the reference's DetrResNetBackbone on the CPU (oracle/make_golden_r03.py; reference
    detr_resnet.py:256-340).  fp32: output 1e-3, BN running statistics 1e-3, gradient norms 1e-2, samples
    max(2e-2, 2 x the reference's own reorder noise); bf16: against the reference's own bf16-autocast deviation (stored)."""
    from simpleaicv_pytorch_training_examples_amd import ops
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.backbones import detr_resnet
    fx = load_golden('detr_r50_stem_layer1_1333')
    torch.manual_seed(fx['model_seed'])
    m = detr_resnet.detr_resnet50backbone().cuda().train()
    g = torch.Generator().manual_seed(fx['data_seed'])
    x = torch.randn(1, 3, fx['h'], fx['w'], generator=g)
    probe = torch.randn(fx['output_shape'], generator=g)
    assert abs(float(x.double().sum()) - fx['input_checksum']) < 1e-6
    x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).cuda()      # NHWC memory, as the collater hands it over
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        # DetrResNetBackbone.forward up to C2 (detr_resnet.py: pack_stem_input -> conv1 -> max_pool2d -> layer1)
        xs = m.conv1(ops.pack_stem_input(x, m.conv1.layer[0]))
        out = m.layer1(ops.max_pool2d(xs, m.maxpool1.kernel_size, m.maxpool1.stride, m.maxpool1.padding))
    assert list(out.shape) == fx['output_shape']
    (out.float() * probe.cuda()).sum().backward()
    torch.cuda.synchronize()
    otol = 1e-3 if dtype == torch.float32 else 1.5 * fx['reference_noise']['bf16_output'] + 2e-2
    assert rel_err(out.float()[:, :, ::8, ::8], fx['output_sub']) < otol
    assert rel_err(out.float()[:, :, 101, :], fx['output_row']) < otol * float(fx['output_sub'].abs().max() / fx['output_row'].abs().max())
    assert abs(float(out.float().norm()) - fx['output_norm']) < (1e-3 if dtype == torch.float32 else 1e-2) * fx['output_norm']
    params = dict(m.named_parameters())
    if dtype == torch.float32:
        gtol = max(2e-2, 2 * fx['reference_noise']['fp32_reorder_grad_sample'])
        worst = 0.0
        for n in fx['used_params']:
            p = params[n]
            assert p.grad is not None, n
            ref_n = fx['grad_norm'][n]
            assert abs(float(p.grad.norm()) - ref_n) <= 1e-2 * max(ref_n, 1e-6), (n, float(p.grad.norm()), ref_n)
            e = rel_err(p.grad.flatten()[:64], fx['grad_sample'][n])
            worst = max(worst, e)
            assert e < gtol, (n, e, gtol)
            if n in fx['grad_full']:
                assert rel_err(p.grad, fx['grad_full'][n]) < gtol, n
        for n, b in m.named_buffers():
            if n in fx['buffers_after'] and b.dtype.is_floating_point:
                assert rel_err(b, fx['buffers_after'][n]) < 1e-3, n
        print(f'detr_r50_stem_layer1_1333 fp32: worst gradient-sample error {worst:.2e} (gate {gtol:.2e})')
    else:
        a = torch.cat([params[n].grad.flatten()[:64].double().cpu() for n in fx['used_params']])
        b = torch.cat([fx['grad_sample'][n].double() for n in fx['used_params']])
        cos = float(a @ b / (a.norm() * b.norm()))
        noise = fx['reference_noise']
        print(f'detr_r50_stem_layer1_1333 bf16: gradient-sample cosine {cos:.4f} (reference bf16 autocast {noise["bf16_grad_sample_cos"]:.4f})')
        assert cos > noise['bf16_grad_sample_cos'] - 0.1
