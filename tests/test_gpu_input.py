"""On-device batch preparation (csrc/input.hip) on a real MI355X.

saicv_mixup_cutmix / saicv_soft_labels : MixupCutmixClassificationCollater.apply_on_device against the fixture the REFERENCE
    collater produced (tests/golden/mixup_cutmix.pt, oracle/make_golden_mixup.py): same numpy seed -> bit-identical images and
    soft labels in all three modes; the uint8 + normalisation form against the same expressions in torch.
saicv_sam_sample_point : the click of reference tools/interactive_segmentation_scripts.py:202-228 -- always inside the error
    region with the label of its kind, a background pixel when the prediction is exact, uniform over the candidate
    (pixel, label) slots (chi-square over 6000 draws)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.make_golden_mixup import batch

pytestmark = pytest.mark.gpu


def test_device_mixup_cutmix_is_bit_identical_to_the_reference_collater():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import MixupCutmixClassificationCollater
    for case in load_golden('mixup_cutmix'):
        np.random.seed(case['np_seed'])
        data = batch(case['data_seed'])
        images = torch.from_numpy(np.array([s['image'] for s in data]).astype(np.float32)).cuda()          # [B, H, W, C]
        labels = torch.tensor([s['label'] for s in data], dtype=torch.int64).cuda()
        got = MixupCutmixClassificationCollater(num_classes=10, **case['kwargs']).apply_on_device(images, labels)
        torch.cuda.synchronize()
        assert got['image'].shape == case['image'].shape and got['image'].stride()[1] == 1      # NHWC-strided NCHW view
        assert torch.equal(got['image'].cpu(), case['image']), case['kwargs']
        assert float((got['label'].cpu() - case['label']).abs().max()) == 0.0, case['kwargs']


def test_device_mixup_cutmix_from_uint8_with_normalisation():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import MixupCutmixClassificationCollater
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (8, 20, 24, 3), generator=g, dtype=torch.uint8)
    labels = torch.randint(0, 10, (8,), generator=g)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    scale, shift = 1.0 / (255.0 * std), -mean / std
    for mode in ('batch', 'pair', 'elem'):
        for seed in (0, 1, 2):
            col = MixupCutmixClassificationCollater(num_classes=10, mode=mode)
            np.random.seed(seed)
            got = col.apply_on_device(u8.cuda(), labels.cuda(), scale.cuda(), shift.cuda())
            np.random.seed(seed)
            plan = col.plan(8, 20, 24)
            x = u8.float() * scale + shift                      # two rounded operations, as the kernel does them
            ref = x.clone()
            for i, (m, yl, yh, xl, xh, lam, oml, _, _) in enumerate(plan):
                j = 7 - i
                if m == 1:
                    ref[i] = x[i] * torch.tensor(lam) + x[j] * torch.tensor(oml)
                elif m == 2:
                    ref[i, yl:yh, xl:xh] = x[j, yl:yh, xl:xh]
            torch.cuda.synchronize()
            assert torch.equal(got['image'].permute(0, 2, 3, 1).cpu(), ref), (mode, seed)
            assert abs(float(got['label'].sum()) - 8.0) < 1e-4


def _click(gt, logits=None, channel=None, seed=0, thr=0.0):
    from simpleaicv_pytorch_training_examples_amd.tools.interactive_segmentation_scripts import sample_error_click
    return sample_error_click(gt, logits, channel, 0.5, thr, seed=seed)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_sam_click_lies_in_the_error_region_with_the_label_of_its_kind(dtype):
    g = torch.Generator().manual_seed(11)
    b, m, h, w = 5, 3, 40, 56
    gt = (torch.rand(b, 1, h, w, generator=g) > 0.6).float().cuda()
    logits = torch.randn(b, m, h, w, generator=g).cuda().to(dtype)
    ch = torch.randint(0, m, (b,), generator=g).cuda()
    pred = logits[torch.arange(b), ch].float() > 0.0
    gtb = gt[:, 0] > 0.5
    for seed in range(40):
        pts = _click(gt, logits, ch, seed)
        torch.cuda.synchronize()
        assert pts.shape == (b, 1, 3)
        for i in range(b):
            x, y, lab = int(pts[i, 0, 0]), int(pts[i, 0, 1]), int(pts[i, 0, 2])
            assert 0 <= x < w and 0 <= y < h
            if lab == 1:
                assert bool(gtb[i, y, x]) and not bool(pred[i, y, x])        # a missed foreground pixel
            else:
                assert not bool(gtb[i, y, x]) and bool(pred[i, y, x])        # a falsely predicted one
    # exact prediction: a background pixel with label 0; nothing predicted at all (pred None): only misses can be clicked
    exact = torch.where(gt[:, :1] > 0.5, 5.0, -5.0).to(dtype)
    pts = _click(gt, exact, None, 3)
    for i in range(b):
        x, y, lab = int(pts[i, 0, 0]), int(pts[i, 0, 1]), int(pts[i, 0, 2])
        assert lab == 0 and not bool(gtb[i, y, x])
    pts = _click(gt, None, None, 4)
    for i in range(b):
        x, y, lab = int(pts[i, 0, 0]), int(pts[i, 0, 1]), int(pts[i, 0, 2])
        assert lab == 1 and bool(gtb[i, y, x])
    full = torch.ones(2, 1, 8, 8).cuda()
    pts = _click(full, torch.full((2, 1, 8, 8), 3.0).cuda(), None, 5)          # everything foreground and predicted: pixel 0, label 0
    assert torch.equal(pts.cpu(), torch.zeros(2, 1, 3))


def test_sam_click_is_uniform_over_the_candidate_slots():
    # 3 false-positive pixels (label 0) and 5 false-negative ones (label 1) in a 16 x 16 mask: 8 slots, 6000 seeded draws
    gt = torch.zeros(1, 1, 16, 16)
    logit = torch.full((1, 1, 16, 16), -1.0)
    fp = [(2, 3), (9, 9), (15, 0)]
    fn = [(0, 0), (4, 7), (8, 8), (12, 1), (13, 14)]
    for y, x in fp:
        logit[0, 0, y, x] = 1.0
    for y, x in fn:
        gt[0, 0, y, x] = 1.0
    gt, logit = gt.cuda(), logit.cuda()
    n = 6000
    pts = torch.cat([_click(gt, logit, None, seed) for seed in range(n)], 0).cpu()
    counts = {}
    for x, y, lab in pts[:, 0].tolist():
        counts[(int(y), int(x), int(lab))] = counts.get((int(y), int(x), int(lab)), 0) + 1
    assert set(counts) == {(y, x, 0) for y, x in fp} | {(y, x, 1) for y, x in fn}
    exp = n / 8
    chi2 = sum((c - exp) ** 2 / exp for c in counts.values())
    assert chi2 < 29.9, (chi2, counts)            # chi-square, 7 degrees of freedom: p = 1e-4


# ------------------------------------------------------------------------------------------ SAM prompt tokens (f3 c)
def _tokens_ref(points, boxes, gauss, table, image_size, pad):
    """fp64 restatement of reference prompt_encoder.py:28-49,150-190: encoding of the pixel centre, then per kind the learned
    row is added (clicks 0 / 1, corners) or replaces the encoding (label -1)."""
    import math
    toks = []
    def enc(xy):
        c = 2 * (xy / image_size) - 1
        ph = 2 * math.pi * (c.double() @ gauss.double())
        return torch.cat([ph.sin(), ph.cos()], -1)
    if points is not None:
        xy, lab = points[:, :, :2] + 0.5, points[:, :, 2]
        if pad:
            xy = torch.cat([xy, torch.zeros(xy.shape[0], 1, 2)], 1)
            lab = torch.cat([lab, -torch.ones(lab.shape[0], 1)], 1)
        e = enc(xy)
        e[lab == -1] = 0
        e[lab == -1] += table[4].double()
        e[lab == 0] += table[0].double()
        e[lab == 1] += table[1].double()
        toks.append(e)
    if boxes is not None:
        e = enc(boxes.reshape(-1, 2, 2) + 0.5)
        e[:, 0] += table[2].double()
        e[:, 1] += table[3].double()
        toks.append(e)
    return torch.cat(toks, 1)


@pytest.mark.parametrize('have_points,have_boxes', [(True, False), (True, True), (False, True)])
def test_sam_prompt_tokens_match_the_formula_and_route_gradients(have_points, have_boxes):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything.prompt_encoder import PromptEncoder
    torch.manual_seed(5)
    enc = PromptEncoder(image_size=1024, patch_size=16, embedding_planes=256, mask_inter_planes=16).cuda()
    g = torch.Generator().manual_seed(6)
    b = 5
    points = boxes = None
    if have_points:
        points = torch.cat([torch.rand(b, 3, 2, generator=g) * 1023, torch.tensor([[1., 0., -1.]]).expand(b, 3).unsqueeze(-1)], -1)
        points[2, 1, 2] = 2.0                                   # a label the tables do not know: encoding only
    if have_boxes:
        lo = torch.rand(b, 2, generator=g) * 500
        boxes = torch.cat([lo, lo + 100 + torch.rand(b, 2, generator=g) * 400], -1)
    sparse, dense = enc(points.cuda() if have_points else None, boxes.cuda() if have_boxes else None, None)
    table = torch.cat([e.weight for e in enc.point_embeddings] + [enc.not_a_point_embed.weight]).detach().cpu()
    ref = _tokens_ref(points, boxes, enc.pe_layer.positional_encoding_gaussian_matrix.cpu(), table, 1024, pad=not have_boxes)
    assert tuple(sparse.shape) == tuple(ref.shape) and sparse.dtype == torch.float32
    assert float((sparse.cpu().double() - ref).abs().max()) < 2e-5        # fp32 phases up to ~25 rad
    assert tuple(dense.shape) == (b, 256, 64, 64)
    # backward: every token's gradient lands in the row of its kind, nowhere else
    w = torch.randn(sparse.shape, generator=g).cuda()
    (sparse * w).sum().backward()
    kinds = []
    if have_points:
        lab = points[:, :, 2]
        if not have_boxes:
            lab = torch.cat([lab, -torch.ones(b, 1)], 1)
        kinds.append(torch.where(lab == -1, 4, torch.where(lab == 0, 0, torch.where(lab == 1, 1, 5))))
    if have_boxes:
        kinds.append(torch.tensor([[2, 3]]).expand(b, 2))
    kinds = torch.cat(kinds, 1)
    rows = [e.weight for e in enc.point_embeddings] + [enc.not_a_point_embed.weight]
    for k, p in enumerate(rows):
        want = (w.cpu() * (kinds == k).unsqueeze(-1)).sum((0, 1))
        got = p.grad.cpu().flatten() if p.grad is not None else torch.zeros(256)
        assert float((got - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max())), k
    # the dense position encoding is the same formula on the grid centres
    pe = enc.get_dense_pe_layer()[0].cpu()
    ij = (torch.arange(64).float() + 0.5) / 64
    grid = torch.stack([ij.view(1, 64).expand(64, 64), ij.view(64, 1).expand(64, 64)], -1) * 1024 - 0.5
    want = _tokens_ref(torch.cat([grid.reshape(1, -1, 2), torch.full((1, 4096, 1), 2.0)], -1), None,
                       enc.pe_layer.positional_encoding_gaussian_matrix.cpu(), table, 1024, pad=False)
    assert float((pe.permute(1, 2, 0).reshape(4096, 256).double() - want[0]).abs().max()) < 2e-5


def test_detr_sine_position_embedding_matches_the_formula():
    """saicv_detr_sine_pe against an fp64 restatement of reference detr_resnet.py:37-64 (two cumulative sums of the un-padded mask,
    normalised to 2 pi, divided by temperature^(2 (k // 2) / F), sin on even / cos on odd features, row features first)."""
    import math
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.backbones.detr_resnet import PositionEmbeddingBlock
    g = torch.Generator().manual_seed(9)
    b, h, w, f = 3, 25, 42, 128
    masks = torch.ones(b, h, w, dtype=torch.bool)
    for i, (hh, ww) in enumerate([(25, 42), (19, 30), (7, 41)]):
        masks[i, :hh, :ww] = False
    masks[1, 3, 5] = True                                     # a hole: the counts skip it
    pe = PositionEmbeddingBlock(inplanes=f).cuda()(masks.cuda()).cpu()
    nm = (~masks).double()
    ye, xe = nm.cumsum(1), nm.cumsum(2)
    ye = ye / (ye[:, -1:, :] + 1e-6) * 2 * math.pi
    xe = xe / (xe[:, :, -1:] + 1e-6) * 2 * math.pi
    k = torch.arange(f, dtype=torch.float64)
    dim_t = 10000 ** (2 * (k // 2) / f)
    def feats(e):
        v = e[..., None] / dim_t
        return torch.where(k % 2 == 0, v.sin(), v.cos())
    ref = torch.cat([feats(ye), feats(xe)], -1).permute(0, 3, 1, 2)
    assert tuple(pe.shape) == (b, 2 * f, h, w) and pe.dtype == torch.float32
    assert float((pe.double() - ref).abs().max()) < 2e-5


# ------------------------------------------------------------------------------------------------ uint8 batch -> normalised fp32
def test_uint8_batch_normalised_on_the_device_equals_the_host_transform():
    """Uint8ClassificationCollater + normalize_on_device against torchvision's ToTensor + Normalize as the reference's
    TorchMeanStdNormalize applies them per sample (classification/common.py:228-248; torchvision is not in the image: its two
    steps are `img.to(float32).div(255)` and `tensor.sub_(mean).div_(std)`, restated here with torch ops).  Bit-identical, and
    the result is the NHWC-strided [B, 3, H, W] view ClassificationCollater hands to the loop."""
    import numpy as np
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import (ClassificationCollater, Uint8ClassificationCollater,
                                                                                       normalize_on_device)
    rng = np.random.default_rng(5)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    samples = [{'image': rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8), 'label': int(rng.integers(0, 10))} for _ in range(5)]
    batch = Uint8ClassificationCollater()(samples)
    assert batch['image'].dtype == torch.uint8 and tuple(batch['image'].shape) == (5, 37, 53, 3)
    out = normalize_on_device(batch['image'].cuda(), mean, std)
    m, s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    host = []
    for smp in samples:                                              # ToTensor, Normalize, permute back to HWC (the reference transform)
        t = torch.from_numpy(smp['image']).permute(2, 0, 1).to(torch.float32).div(255)
        host.append({'image': t.sub_(m).div_(s).permute(1, 2, 0).numpy(), 'label': smp['label']})
    ref = ClassificationCollater()(host)
    assert tuple(out.shape) == tuple(ref['image'].shape) and out.stride() == ref['image'].stride()
    assert torch.equal(out.cpu(), ref['image'])
    assert torch.equal(batch['label'], ref['label'])


@pytest.mark.parametrize('mode', ['const', 'rand', 'pixel'])
def test_random_erasing_on_the_device_batch(mode):
    """RandomErasing.plan + erase_on_device on a [B, H, W, C] device batch against the reference's per-sample host call
    (tests/golden/random_erasing.pt): 'const' / 'rand' bit-identical (boxes AND colours come from the same numpy draws); 'pixel':
    the same rectangle, nothing outside it touched, N(0, 1) values inside (the reference draws them on the host: same
    distribution, different generator), deterministic in the seed."""
    import numpy as np
    from conftest import load_golden
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import RandomErasing
    cases = [c for c in load_golden('random_erasing')['cases'] if c['kwargs']['mode'] == mode]
    images, plans, befores = [], [], []
    for c in cases:
        np.random.seed(c['seed'])
        image = np.random.standard_normal((40, 48, 3)).astype(np.float32)
        befores.append(image.copy())
        images.append(image)
        plans.append(RandomErasing(**c['kwargs']).plan(40, 48, 3))           # continues the seeded stream like the host call
    eraser = RandomErasing(prob=1.0, mode=mode)
    x = torch.from_numpy(np.stack(images)).cuda()
    out = eraser.erase_on_device(x, plans, seed=11).cpu()
    again = eraser.erase_on_device(torch.from_numpy(np.stack(befores)).cuda(), plans, seed=11).cpu()
    assert torch.equal(out, again)
    filled = []
    for i, c in enumerate(cases):
        if mode == 'pixel' and 'max_count' in c['kwargs']:
            # several boxes: the host call's later boxes follow the fill draws of the earlier ones, the plan's do not -- only the
            # first box is common to both
            boxes = plans[i][:1]
        else:
            boxes = plans[i]
        if mode != 'pixel':
            assert torch.equal(out[i], c['image']), (c['kwargs'], c['seed'])
            continue
        mask = np.zeros((40, 48), dtype=bool)
        for top, left, h, w, _ in plans[i]:
            mask[top:top + h, left:left + w] = True
        assert torch.equal(out[i][~torch.from_numpy(mask)], torch.from_numpy(befores[i])[~torch.from_numpy(mask)])
        if boxes:
            top, left, h, w, _ = boxes[0]
            ref_changed = (c['image'].numpy() != befores[i]).any(axis=-1)
            assert ref_changed[top:top + h, left:left + w].mean() > 0.97       # the reference erased that same rectangle
            filled.append(out[i][torch.from_numpy(mask)].flatten())
    if mode == 'pixel':
        v = torch.cat(filled).double()
        assert v.numel() > 3000 and abs(float(v.mean())) < 0.06 and abs(float(v.std()) - 1.0) < 0.05
        assert float(v.abs().max()) < 6.0 and float((v.abs() > 2).double().mean()) > 0.02
