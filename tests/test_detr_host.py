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


def _loss_case(seed=0, L=6, B=3, Q=20, C=20):
    import torch
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(L, B, Q, C + 1, generator=g, requires_grad=True)
    reg = torch.rand(L, B, Q, 4, generator=g, requires_grad=True)
    ann = -torch.ones(B, 10, 5)
    for b in range(B):
        n = (2 + b) if b != 1 else 0                       # one image without boxes
        if n:
            ann[b, :n, :2] = torch.rand(n, 2, generator=g) * 0.5 + 0.25
            ann[b, :n, 2:4] = torch.rand(n, 2, generator=g) * 0.3 + 0.1
            ann[b, :n, 4] = torch.randint(0, C, (n,), generator=g).float()
    return cls, reg, ann, C


def test_detr_loss_over_all_layers_equals_the_per_layer_reference_form():
    """DETRLoss.forward computes the six decoder layers in one pass; the reference-named per-layer functions
    (compute_batch_cls_loss / compute_batch_l1_iou_loss, reference losses.py:905-935) are the yardstick: same 18 terms,
    same gradients."""
    import torch
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    cls, reg, ann, C = _loss_case()
    crit = DETRLoss(num_classes=C)
    got = crit([cls, reg], ann)
    sum(got.values()).backward()
    g_cls, g_reg = cls.grad.clone(), reg.grad.clone()
    cls.grad = reg.grad = None
    regc = torch.clamp(reg, 1e-4, 1 - 1e-4)
    idx = crit.get_matched_pred_target_idxs(cls[-1].detach(), regc[-1].detach(), ann)
    ref = {}
    for i in range(cls.shape[0]):
        ref[f'layer_{i}_cls_loss'] = crit.cls_loss_weight * crit.compute_batch_cls_loss(cls[i], ann, idx)
        l1, iou = crit.compute_batch_l1_iou_loss(regc[i], ann, idx)
        ref[f'layer_{i}_box_l1_loss'] = crit.box_l1_loss_weight * l1
        ref[f'layer_{i}_box_iou_loss'] = crit.iou_loss_weight * iou
    sum(ref.values()).backward()
    assert set(got) == set(ref) and len(got) == 18
    for k in ref:
        assert abs(float(got[k]) - float(ref[k])) < 1e-5 * max(1.0, abs(float(ref[k]))), k
    assert float((g_cls - cls.grad).abs().max()) < 1e-6 and float((g_reg - reg.grad).abs().max()) < 1e-6


def test_detr_loss_selects_valid_rows_on_the_host_copy_of_the_annotations():
    """With `annotations._saicv_host` attached (train_detection does that) the valid-row selection needs no device
    synchronisation; the loss terms are the same numbers as through the boolean-mask path."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    cls, reg, ann, C = _loss_case(seed=3)
    crit = DETRLoss(num_classes=C)
    plain = crit([cls.detach(), reg.detach()], ann.clone())
    tagged = ann.clone()
    tagged._saicv_host = ann.clone()
    fast = crit([cls.detach(), reg.detach()], tagged)
    assert all(float(plain[k]) == float(fast[k]) for k in plain)


def test_detr_static_shape_loss_equals_the_dynamic_loss():
    """DETRLoss.match_inputs -> assign_host -> forward_static (r05: the form the captured step uses, with the assignment from scipy on
    the host here) against DETRLoss.forward on the same predictions and annotations: every loss term to 1e-6 and the gradients with
    respect to both prediction tensors to 1e-6 of their scale.  Images with 0, 1 and several boxes, padding rows INTERLEAVED with
    boxes, the ground truth padded to 8 rows (the static buffer) against the collater's 100-row tensor on the dynamic side."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    g = torch.Generator().manual_seed(21)
    layers, b, q, classes, t = 3, 4, 12, 20, 8
    cls_preds = torch.randn(layers, b, q, classes + 1, generator=g)
    reg_preds = torch.rand(layers, b, q, 4, generator=g) * 0.8 + 0.1
    annots = -torch.ones(b, t, 5)
    rows = {0: [1, 4, 6], 1: [], 2: [0], 3: [0, 1, 2, 3, 5]}
    for i, rr in rows.items():
        for r in rr:
            cxcy = torch.rand(2, generator=g) * 0.5 + 0.25
            wh = torch.rand(2, generator=g) * 0.3 + 0.1
            annots[i, r] = torch.cat([cxcy, wh, torch.randint(0, classes, (1,), generator=g).float()])
    crit = DETRLoss(num_classes=classes)

    def run(static):
        c, r = cls_preds.clone().requires_grad_(True), reg_preds.clone().requires_grad_(True)
        if static:
            cost, valid = crit.match_inputs([c, r], annots)
            src, tgt, w = crit.assign_host(cost, valid)
            assert int(w.sum()) == sum(len(v) for v in rows.values())
            ld = crit.forward_static([c, r], annots, src, tgt, w)
        else:
            wide = -torch.ones(b, 100, 5)
            wide[:, :t] = annots
            ld = crit([c, r], wide)
        sum(ld.values()).backward()
        return {k: float(v) for k, v in ld.items()}, c.grad, r.grad

    dyn, dc, dr = run(False)
    sta, sc, sr = run(True)
    assert list(dyn.keys()) == list(sta.keys()) and len(dyn) == 3 * layers
    for k in dyn:
        assert abs(dyn[k] - sta[k]) <= 1e-6 * max(1.0, abs(dyn[k])), (k, dyn[k], sta[k])
    assert float((dc - sc).abs().max()) <= 1e-6 * float(dc.abs().max())
    assert float((dr - sr).abs().max()) <= 1e-6 * float(dr.abs().max())


def test_detr_static_shape_loss_divides_by_the_boxes_of_the_batch():
    """forward_static normalised the box losses by the number of MATCHED pairs, clamped at 1.  The reference
    (detection/losses.py:938-954) and forward() divide by the number of ground-truth boxes, unclamped: (a) an image with more boxes
    than queries matches Q pairs but divides by its n boxes -- static and dynamic forms must agree term by term; (b) a batch without
    any box gives 0 / 0 = nan in both forms (tools.scripts then skips the step), where the clamp trained a no-object step."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    g = torch.Generator().manual_seed(5)
    layers, b, q, classes, t = 2, 2, 3, 20, 6
    cls_preds = torch.randn(layers, b, q, classes + 1, generator=g)
    reg_preds = torch.rand(layers, b, q, 4, generator=g) * 0.8 + 0.1
    crit = DETRLoss(num_classes=classes)

    def annots_with(rows):
        a = -torch.ones(b, t, 5)
        for i, n in enumerate(rows):
            for r in range(n):
                a[i, r] = torch.cat([torch.rand(2, generator=g) * 0.5 + 0.25, torch.rand(2, generator=g) * 0.3 + 0.1,
                                     torch.randint(0, classes, (1,), generator=g).float()])
        return a

    def both(annots):
        cost, valid = crit.match_inputs([cls_preds, reg_preds], annots)
        src, tgt, w = crit.assign_host(cost, valid)
        sta = crit.forward_static([cls_preds, reg_preds], annots, src, tgt, w)
        dyn = crit([cls_preds, reg_preds], annots)
        return {k: float(v) for k, v in dyn.items()}, {k: float(v) for k, v in sta.items()}, int(w.sum())

    dyn, sta, pairs = both(annots_with([5, 1]))          # image 0: five boxes, three queries
    assert pairs == 3 + 1
    for k in dyn:
        assert abs(dyn[k] - sta[k]) <= 1e-6 * max(1.0, abs(dyn[k])), (k, dyn[k], sta[k])
    dyn, sta, pairs = both(annots_with([0, 0]))
    assert pairs == 0
    for k in dyn:
        if 'box' in k:
            assert np.isnan(dyn[k]) and np.isnan(sta[k]), (k, dyn[k], sta[k])
        else:
            assert abs(dyn[k] - sta[k]) <= 1e-6 * max(1.0, abs(dyn[k])), (k, dyn[k], sta[k])


def test_layer_terms_are_the_weighted_selects_and_their_gradients():
    """DETRLoss._layer_terms (late r06): the reference's dict of 3 L weighted scalars (losses.py:905-935) as unbind() views of three
    weighted vectors -- the same values and the same gradients as `weight * vector[idx]` entry by entry, bit for bit."""
    import torch
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
    crit = DETRLoss(cls_loss_weight=1.0, box_l1_loss_weight=5.0, iou_loss_weight=2.0)
    g = torch.Generator().manual_seed(3)
    vecs = [torch.randn(6, generator=g).requires_grad_(True) for _ in range(3)]
    ref_vecs = [v.detach().clone().requires_grad_(True) for v in vecs]
    terms = crit._layer_terms(*vecs)
    ref = {}
    for idx in range(6):
        ref[f'layer_{idx}_cls_loss'] = 1.0 * ref_vecs[0][idx]
        ref[f'layer_{idx}_box_l1_loss'] = 5.0 * ref_vecs[1][idx]
        ref[f'layer_{idx}_box_iou_loss'] = 2.0 * ref_vecs[2][idx]
    assert list(terms) == list(ref)
    coeff = torch.randn(18, generator=g)
    torch.stack(list(terms.values())).mul(coeff).sum().backward()
    torch.stack(list(ref.values())).mul(coeff).sum().backward()
    for k in ref:
        assert torch.equal(terms[k], ref[k]), k
    for a, b in zip(vecs, ref_vecs):
        assert torch.equal(a.grad, b.grad)
