"""DETR host-side pieces on CPU: the set-prediction loss against the fixture produced by the reference's own
DETRLoss (same outputs in -> same 18 loss terms and the same Hungarian assignment), the collater contract,
and the drop-in state_dict / init contract of the model."""
import numpy as np
import torch

from conftest import load_golden
from oracle.make_golden_detr import detr_inputs


def test_detr_loss_matches_reference_on_reference_outputs():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    fx = load_golden('detr_r18_tiny')
    crit = DETRLoss(num_classes=fx['kwargs']['num_classes'])
    _, _, annots = detr_inputs(fx['batch'], fx['data_seed'], num_classes=fx['kwargs']['num_classes'])
    cls_out = fx['cls_outputs'].clone().requires_grad_(True)
    reg_out = fx['reg_outputs'].clone().requires_grad_(True)
    ld = crit([cls_out, reg_out], annots)
    assert list(ld.keys()) == list(fx['loss'].keys())
    for k, v in fx['loss'].items():
        assert abs(float(ld[k]) - v) < 1e-5 * max(1.0, abs(v)), (k, float(ld[k]), v)
    idx = crit.get_matched_pred_target_idxs(cls_out[-1].detach(), torch.clamp(reg_out[-1].detach(), 1e-4, 1 - 1e-4), annots)
    for (i, j), (ri, rj) in zip(idx, fx['indices']):
        assert torch.equal(i, ri) and torch.equal(j, rj)
    sum(ld.values()).backward()
    assert torch.isfinite(cls_out.grad).all() and float(reg_out.grad.abs().sum()) > 0


def test_detr_collater_contract():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.common import DETRDetectionCollater
    rng = np.random.RandomState(0)
    data = []
    for h, w, n in ((96, 128, 2), (128, 64, 0)):
        a = np.zeros((n, 5), dtype=np.float32)
        for k in range(n):
            a[k] = [10 + k, 20, 50 + k, 60, k]
        data.append({'image': rng.randn(h, w, 3).astype(np.float32), 'annots': a, 'scale': 0.5, 'size': [2 * h, 2 * w]})
    out = DETRDetectionCollater(resize=128, resize_type='yolo_style', max_annots_num=7)(data)
    img = out['image']
    assert img.shape == (2, 3, 128, 128) and img.stride() == (128 * 128 * 3, 1, 128 * 3, 3)      # NHWC memory
    assert torch.equal(img[0, :, :96, :128], torch.from_numpy(data[0]['image']).permute(2, 0, 1))
    assert out['mask'].dtype == torch.bool and not out['mask'][0, :96, :128].any() and out['mask'][0, 96:, :].all()
    assert out['mask'][1, :, 64:].all() and not out['mask'][1, :128, :64].any()
    assert out['annots'].shape == (2, 7, 5) and float(out['annots'][0, 2:].max()) == -1 and float(out['annots'][1].max()) == -1
    sa = out['scaled_annots'][0, 0]
    assert torch.allclose(sa, torch.tensor([30 / 128, 40 / 96, 40 / 128, 40 / 96, 0.0]))
    assert out['scaled_size'].tolist() == [[96, 128], [128, 64]]
    assert DETRDetectionCollater(resize=800, resize_type='retina_style').resize == 1333


def test_detr_state_dict_and_init_contract():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import detr
    fx = load_golden('detr_r18_tiny')
    torch.manual_seed(fx['model_seed'])
    m = detr.__dict__[fx['factory']](**fx['kwargs'])
    keys = list(m.state_dict().keys())
    assert set(fx['grad_norm'].keys()) == {n for n, _ in m.named_parameters()}
    assert 'transformer.encoder_blocks.0.attention.in_proj_weight' in keys
    assert m.state_dict()['transformer.decoder_blocks.5.multihead_attention.in_proj_weight'].shape == (768, 256)
    assert m.state_dict()['head.cls_head.weight'].shape == (21, 256) and m.state_dict()['query_embed.weight'].shape == (20, 256)
    assert keys[0] == 'backbone.conv1.layer.0.weight' and keys[-1] == 'head.reg_head.4.bias'
