"""RetinaLoss on csrc/detloss.hip (anchor assignment, focal loss, SmoothL1) through the C-ABI against fixtures the REFERENCE produced
(oracle/make_golden_retinaloss.py runs SimpleAICV/detection/losses.py:123-433 on the CPU in fp32).  The class targets of
get_batch_anchors_annotations are integers and must match exactly (every anchor of three images, one of them without ground truth);
loss values within 1e-4, gradient norms within 1e-4 and gradient samples within 1e-4 of the tensor's gradient scale (fp32 on both
sides, another summation order).  box_loss_type SmoothL1 (kernel), GIoU / CIoU (tensor arithmetic on the positive anchors)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _inputs():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    try:
        import make_golden_retinaloss as m
    finally:
        sys.path.pop(0)
    return m.inputs(), m.sample_idx


@pytest.mark.parametrize('case', ['smoothl1', 'giou', 'ciou', 'smoothl1_gamma15'])
def test_retina_loss_matches_reference(case):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import RetinaLoss
    fx = torch.load(os.path.join(GOLD, 'retina_loss.pt'), weights_only=True)[case]
    (cls, reg, annots), sample_idx = _inputs()
    leaves = [t.cuda().requires_grad_(True) for t in cls + reg]
    crit = RetinaLoss(**fx['config'])
    out = crit([leaves[:5], leaves[5:]], annots.cuda())
    assert set(out) == {'cls_loss', 'reg_loss'}
    assert abs(float(out['cls_loss']) - fx['cls_loss']) <= 1e-4 * abs(fx['cls_loss'])
    assert abs(float(out['reg_loss']) - fx['reg_loss']) <= 1e-4 * abs(fx['reg_loss'])
    (out['cls_loss'] + out['reg_loss']).backward()
    for i, t in enumerate(leaves):
        g = t.grad.float().cpu()
        assert abs(float(g.norm()) - fx['grad_norm'][i]) <= 1e-4 * fx['grad_norm'][i] + 1e-9, (i, float(g.norm()), fx['grad_norm'][i])
        diff = (g.flatten()[sample_idx(g.numel())] - fx['grad_sample'][i]).abs().max()
        assert float(diff) <= 1e-4 * float(g.abs().max()) + 1e-9, i


def test_anchor_assignment_is_exact():
    """saicv_retina_assign directly: class targets of all 3 x 3852 anchors equal the reference's, box targets on a sample"""
    from simpleaicv_pytorch_training_examples_amd import _lib
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.anchor import RetinaAnchors
    fx = torch.load(os.path.join(GOLD, 'retina_loss.pt'), weights_only=True)['smoothl1']
    (cls, reg, annots), sample_idx = _inputs()
    sizes = [[t.shape[2], t.shape[1]] for t in cls]
    table = torch.cat([torch.from_numpy(a).view(-1, 4) for a in RetinaAnchors()(sizes)], dim=0).cuda()
    a = table.shape[0]
    targets = torch.empty(3, a, 5, device='cuda')
    pos = torch.zeros(1, device='cuda')
    ann = annots.cuda().contiguous()
    _lib.check(_lib.lib().saicv_retina_assign(table.data_ptr(), ann.data_ptr(), targets.data_ptr(), pos.data_ptr(), 3, a, ann.shape[1], 1, None),
               'retina_assign')
    torch.cuda.synchronize()
    assert torch.equal(targets[:, :, 4].cpu().to(torch.int8), fx['class_targets'])
    assert int(pos) == sum(c[2] for c in fx['census'])
    got = targets[0, :, 0:4].cpu()[sample_idx(a, 64)]
    assert float((got - fx['box_targets_sample']).abs().max()) <= 1e-5 * float(fx['box_targets_sample'].abs().max())
    assert bool((targets[1] == -1).all())                     # the image without ground truth: everything ignored


def test_focal_loss_at_a_training_sized_level_is_additive_and_ignores_masked_rows():
    """Size-independent properties at the largest level of a 1024 x 1024 RetinaNet batch (8 x 128 x 128 x 9 anchors x 80 classes,
    94 M elements): the sum over the batch equals the sum over its halves; rows with class target -1 contribute neither loss nor
    gradient; the gradient of clamped probabilities is zero."""
    from simpleaicv_pytorch_training_examples_amd import _lib
    L = _lib.lib()
    b, al, c = 8, 128 * 128 * 9, 80
    g = torch.Generator(device='cuda').manual_seed(0)
    probs = torch.rand(b, al, c, device='cuda', generator=g) * 0.998 + 0.001
    probs[:, ::7, 3] = 2e-5
    targets = torch.zeros(b, al, 5, device='cuda')
    cls = torch.randint(-1, c + 1, (b, al), device='cuda', generator=g).float()
    targets[:, :, 4] = cls
    dp = torch.empty_like(probs)

    def run(p, t, d, n):
        s = torch.zeros(1, device='cuda')
        _lib.check(L.saicv_focal_loss_level(p.data_ptr(), t.data_ptr(), d.data_ptr() if d is not None else None, s.data_ptr(), n, al, al, 0, c,
                                            0.25, 2.0, None), 'focal')
        return float(s)

    full = run(probs, targets, dp, b)
    lo, hi = run(probs[:4], targets[:4], None, 4), run(probs[4:], targets[4:], None, 4)
    assert abs(full - (lo + hi)) <= 1e-4 * abs(full)
    assert float(dp[cls < 0].abs().max()) == 0.
    live = cls >= 0
    assert float(dp[:, ::7, 3][live[:, ::7]].abs().max()) == 0.
    # one row against the torch formulation
    r = (cls[0] > 0).nonzero()[0].item()
    p = probs[0, r].clone().requires_grad_(True)
    y = torch.zeros(c, device='cuda')
    y[int(cls[0, r]) - 1] = 1.
    pc = p.clamp(1e-4, 1 - 1e-4)
    pt = torch.where(y == 1, pc, 1 - pc)
    w = torch.where(y == 1, torch.full_like(pc, 0.25), torch.full_like(pc, 0.75))
    (-(w * (1 - pt) ** 2 * torch.log(pt))).sum().backward()
    assert float((dp[0, r] - p.grad).abs().max()) <= 1e-5 * float(p.grad.abs().max())
