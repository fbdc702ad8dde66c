"""FCOSLoss on csrc/detloss.hip (point assignment, focal loss) + tensor arithmetic on the positive points, against fixtures the
REFERENCE produced (oracle/make_golden_fcosloss.py runs SimpleAICV/detection/losses.py:434-842 on the CPU in fp32).  The class
targets of get_batch_position_annotations are integers and must match exactly on every point of three images (one without ground
truth), with and without centre sampling and for two sets of regression ranges; (l, t, r, b) and centre-ness targets within 1e-6;
the three loss values within 1e-4, gradient norms within 1e-4, gradient samples within 1e-4 of the tensor's gradient scale."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
CASES = ['giou', 'ciou', 'iou_nocenter', 'default_ranges', 'default_ranges_nocenter']


def _inputs():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    try:
        import make_golden_fcosloss as m
    finally:
        sys.path.pop(0)
    return m.inputs(), m.sample_idx


@pytest.mark.parametrize('case', CASES)
def test_fcos_loss_matches_reference(case):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import FCOSLoss
    fx = torch.load(os.path.join(GOLD, 'fcos_loss.pt'), weights_only=True)[case]
    (cls, reg, ctr, annots), sample_idx = _inputs()
    leaves = [t.cuda().requires_grad_(True) for t in cls + reg + ctr]
    crit = FCOSLoss(**fx['config'])
    out = crit([leaves[:5], leaves[5:10], leaves[10:]], annots.cuda())
    assert set(out) == set(fx['losses'])
    for k, v in fx['losses'].items():
        assert abs(float(out[k]) - v) <= 1e-4 * abs(v), (k, float(out[k]), v)
    sum(out.values()).backward()
    for i, t in enumerate(leaves):
        g = t.grad.float().cpu()
        assert abs(float(g.norm()) - fx['grad_norm'][i]) <= 1e-4 * fx['grad_norm'][i] + 1e-9, (i, float(g.norm()), fx['grad_norm'][i])
        diff = (g.flatten()[sample_idx(g.numel())] - fx['grad_sample'][i]).abs().max()
        assert float(diff) <= 1e-4 * float(g.abs().max()) + 1e-9, i


@pytest.mark.parametrize('case', CASES)
def test_point_assignment_is_exact(case):
    from simpleaicv_pytorch_training_examples_amd import _lib
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import FCOSLoss
    fx = torch.load(os.path.join(GOLD, 'fcos_loss.pt'), weights_only=True)[case]
    (cls, reg, ctr, annots), _ = _inputs()
    crit = FCOSLoss(**fx['config'])
    points = crit._point_table([[t.shape[2], t.shape[1]] for t in cls], torch.device('cuda'))
    p = points.shape[0]
    targets = torch.empty(3, p, 5, device='cuda')
    cness = torch.empty(3, p, device='cuda')
    pos = torch.zeros(1, device='cuda')
    ann = annots.cuda().contiguous()
    _lib.check(_lib.lib().saicv_fcos_assign(points.data_ptr(), ann.data_ptr(), targets.data_ptr(), cness.data_ptr(), pos.data_ptr(), 3, p,
                                            ann.shape[1], float(crit.center_sample_radius), int(crit.use_center_sample), None), 'fcos_assign')
    torch.cuda.synchronize()
    assert torch.equal(targets[:, :, 4].cpu().to(torch.int8), fx['class_targets'])
    assert int(pos) == int((fx['class_targets'] > 0).sum())
    assert float((targets[:, :, 0:4].cpu() - fx['ltrb']).abs().max()) <= 1e-6 * float(fx['ltrb'].abs().max())
    assert float((cness.cpu() - fx['centerness']).abs().max()) <= 1e-6
    assert bool((targets[1] == 0).all()) and bool((cness[1] == 0).all())          # the image without ground truth
