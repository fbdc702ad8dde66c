"""Round-4 host-side logic (CPU): the relative-position table resampling against torch's own linear interpolation, and the
ConvBnActBlock switches' parameter contract against the reference-generated fixture (no kernels run here)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden


def test_resize_rel_pos_equals_linear_interpolation_and_its_gradient():
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    g = torch.Generator().manual_seed(0)
    for src, dst in [(27, 127), (127, 27), (27, 63), (63, 64), (13, 127), (15, 31)]:
        t = torch.randn(src, 64, generator=g, requires_grad=True)
        ref = F.interpolate(t.reshape(1, src, -1).permute(0, 2, 1), size=dst, mode='linear').reshape(-1, dst).permute(1, 0)
        out = ops_tfm.resize_rel_pos(t, dst)
        assert out.shape == (dst, 64) and out.is_contiguous() and out.dtype == torch.float32
        assert float((out - ref).abs().max()) < 1e-5
        probe = torch.randn(dst, 64, generator=g)
        (g_ref,) = torch.autograd.grad(ref, t, probe, retain_graph=True)
        (g_out,) = torch.autograd.grad(out, t, probe)
        assert float((g_out - g_ref).abs().max()) < 1e-4
    t = torch.randn(31, 64)
    assert ops_tfm.resize_rel_pos(t, 31) is t                     # the native grid: the parameter itself, no copy


def test_convbnact_block_switches_keep_the_reference_parameter_contract():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import ConvBnActBlock
    cases = load_golden('convbnact_variants')['cases']
    for key, c in cases.items():
        blk = ConvBnActBlock(**c['kwargs'])
        sd = blk.state_dict()
        assert list(sd.keys()) == list(c['state_dict'].keys()), key
        for k in sd:
            assert sd[k].shape == c['state_dict'][k].shape, (key, k)
        blk.load_state_dict(c['state_dict'])


def _vote(world, finish_after, seconds=1.0):
    """Drive bench.WatchdogVote with `world` in-process ranks over one HashStore: rank r calls finished() after
    finish_after[r] seconds (None: never).  -> (decisions, acted)."""
    import threading
    import time
    import torch.distributed as dist
    import bench
    store = dist.HashStore()
    acted = []
    lock = threading.Lock()

    def act(secs, why, r):
        with lock:
            acted.append((r, why))

    votes = [bench.WatchdogVote(store, r, world, seconds, epoch=7, act=lambda s, w, r=r: act(s, w, r), poll=0.02) for r in range(world)]

    def finisher(r):
        if finish_after[r] is not None:
            time.sleep(finish_after[r])
            votes[r].finished()

    ts = [threading.Thread(target=finisher, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for v in votes:
        v.thread.join(seconds + 10)
        assert not v.thread.is_alive()
    return [v.decision for v in votes], sorted(acted)


def test_watchdog_vote_everyone_finished_nobody_reexecutes():
    decisions, acted = _vote(3, [0.05, 0.1, 0.2])
    assert decisions == ['ok', 'ok', 'ok'] and acted == []


def test_watchdog_vote_one_stalled_rank_sends_every_rank_to_eager():
    decisions, acted = _vote(3, [0.05, None, 0.1])
    assert decisions == ['reexec'] * 3
    assert [r for r, _ in acted] == [0, 1, 2] and all(w == 'collective decision' for _, w in acted)


def test_watchdog_vote_is_one_decision_even_when_a_rank_finishes_just_after_the_deadline():
    """The race ADVICE r03 describes: rank 0 done in time, rank 1 a moment too late.  Per-rank timers would send only rank 1 to
    eager; the vote gives every rank the same answer (here: reexec, because rank 0 published before rank 1's key existed)."""
    decisions, acted = _vote(2, [0.1, 1.3], seconds=1.0)
    assert len(set(decisions)) == 1
    assert (decisions[0] == 'reexec') == (len(acted) == 2)


# ------------------------------------------------------------------------------------------------ numpy COCO evaluation
def _gt(img, cat, box, crowd=0):
    return {'image_id': img, 'category_id': cat, 'bbox': list(box), 'area': box[2] * box[3], 'iscrowd': crowd}


def _dt(img, cat, box, score):
    return {'image_id': img, 'category_id': cat, 'bbox': list(box), 'score': score}


def test_cocoeval_perfect_detections_score_one_in_every_populated_cell():
    from simpleaicv_pytorch_training_examples_amd.tools import cocoeval_numpy as CE
    boxes = {0: [(10, 10, 20, 20), (100, 50, 60, 60)], 1: [(5, 5, 200, 150)], 2: [(30, 30, 10, 12)]}
    gts = [_gt(i, c, b) for i, bs in boxes.items() for c, b in enumerate(bs)]
    dts = [_dt(i, c, b, 0.9 - 0.1 * c) for i, bs in boxes.items() for c, b in enumerate(bs)]
    stats, precision, recall = CE.evaluate_bbox(gts, dts)
    assert all(abs(s - 1.0) < 1e-12 for s in stats), stats          # small, medium and large boxes are all present
    assert precision.shape == (10, 101, 2, 4, 3) and recall.shape == (10, 2, 4, 3)


def test_cocoeval_hand_worked_precision_recall_curve():
    """One image, one category, two ground-truth boxes; detections: 0.9 exact on the first, 0.8 nowhere, 0.7 with IoU 0.62 on the
    second.  t <= 0.60: TP FP TP -> recall (.5 .5 1), precision envelope (1 2/3 2/3): AP = (51 + 50 * 2/3) / 101.  t >= 0.65: TP FP FP
    -> AP = 51 / 101.  mAR@100 = (3 * 1 + 7 * 0.5) / 10; with one detection per image only the first box is ever found."""
    from simpleaicv_pytorch_training_examples_amd.tools import cocoeval_numpy as CE
    gts = [_gt(7, 3, (0, 0, 100, 100)), _gt(7, 3, (200, 200, 100, 100))]
    dts = [_dt(7, 3, (0, 0, 100, 100), 0.9), _dt(7, 3, (500, 500, 100, 100), 0.8), _dt(7, 3, (200, 200, 100, 62), 0.7)]
    stats, precision, _ = CE.evaluate_bbox(gts, dts)
    hi, lo = (51 + 50 * 2 / 3) / 101, 51 / 101
    assert abs(stats[1] - hi) < 1e-9                                # IoU = 0.50
    assert abs(stats[2] - lo) < 1e-9                                # IoU = 0.75
    assert abs(stats[0] - (3 * hi + 7 * lo) / 10) < 1e-9
    assert abs(stats[8] - (3 * 1.0 + 7 * 0.5) / 10) < 1e-9          # mAR, 100 detections
    assert abs(stats[6] - 0.5) < 1e-12                              # mAR, 1 detection per image
    assert stats[3] == -1 and stats[4] == -1 and abs(stats[5] - stats[0]) < 1e-12      # both boxes are "large" (area 10 000)
    assert abs(precision[0, 0, 0, 0, 2] - 1.0) < 1e-12 and abs(precision[0, 100, 0, 0, 2] - 2 / 3) < 1e-12      # tp / (tp + fp + spacing(1))


def test_cocoeval_crowd_and_area_rules():
    """A detection on a crowd box is ignored (no TP, no FP) and the crowd box may absorb several; a ground-truth box outside the
    area range is ignored there, and so is an unmatched detection whose own area is outside it."""
    from simpleaicv_pytorch_training_examples_amd.tools import cocoeval_numpy as CE
    gts = [_gt(0, 1, (0, 0, 20, 20)), _gt(0, 1, (300, 300, 200, 200), crowd=1)]
    dts = [_dt(0, 1, (0, 0, 20, 20), 0.9), _dt(0, 1, (310, 310, 50, 50), 0.8), _dt(0, 1, (400, 400, 60, 60), 0.7)]
    stats, _, _ = CE.evaluate_bbox(gts, dts)
    assert abs(stats[0] - 1.0) < 1e-12 and abs(stats[8] - 1.0) < 1e-12      # the two detections inside the crowd box cost nothing
    assert abs(stats[3] - 1.0) < 1e-12 and stats[5] == -1                   # one small regular box; no regular large one
    # without the crowd flag the same two detections are false positives against a second regular box they overlap too little
    gts[1]['iscrowd'] = 0
    stats2, _, _ = CE.evaluate_bbox(gts, dts)
    assert stats2[0] < 0.6 and abs(stats2[8] - 0.5) < 1e-12


def test_cocoeval_is_invariant_under_monotone_score_maps_and_sensitive_to_order():
    import numpy as np
    from simpleaicv_pytorch_training_examples_amd.tools import cocoeval_numpy as CE
    rng = np.random.RandomState(0)
    gts, dts = [], []
    for img in range(6):
        for k in range(4):
            x, y, w, h = rng.rand(4) * np.array([300, 300, 120, 120]) + np.array([0, 0, 8, 8])
            gts.append(_gt(img, k % 2, (x, y, w, h)))
            jit = rng.randn(4) * np.array([4, 4, 6, 6])
            dts.append(_dt(img, k % 2, (x + jit[0], y + jit[1], max(w + jit[2], 2), max(h + jit[3], 2)), float(rng.rand())))
        for _ in range(3):
            dts.append(_dt(img, int(rng.randint(2)), tuple(rng.rand(4) * 300 + 5), float(rng.rand() * 0.5)))
    a, _, _ = CE.evaluate_bbox(gts, dts)
    b, _, _ = CE.evaluate_bbox(gts, [dict(d, score=d['score'] ** 3 + 1.0) for d in dts])
    assert np.allclose(a, b) and 0.05 < a[0] < 1.0
    worse, _, _ = CE.evaluate_bbox(gts, [dict(d, score=1.0 - d['score']) for d in dts])
    assert worse[0] < a[0]


# ------------------------------------------------------------------------------------------------ MAE collater / entry script
def test_mae_collater_reproduces_the_reference_collater_bit_for_bit():
    """MAESelfSupervisedPretrainCollater (reference masked_image_modeling/common.py:16-56) against what the reference class
    returned for the same seeded samples (oracle/make_golden_mae.py): image values AND strides (the loop hands the NHWC-strided
    view to the model), per-patch labels in (p, q, c) order, standardised with the unbiased variance."""
    import numpy as np
    from conftest import load_golden
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.masked_image_modeling.common import MAESelfSupervisedPretrainCollater
    fx = load_golden('mae_collater')['cases']
    assert set(fx) == {'i64_p16_norm', 'i64_p16_raw', 'i32_p8_norm'}
    for key, c in fx.items():
        rng = np.random.default_rng(c['seed'])
        data = [{'image': rng.standard_normal((c['size'], c['size'], 3), dtype=np.float32) * 0.7 + 0.1, 'label': 0} for _ in range(3)]
        out = MAESelfSupervisedPretrainCollater(image_size=c['size'], patch_size=c['patch'], norm_label=c['norm'])(data)
        assert out['image'].dtype == out['label'].dtype == torch.float32
        assert torch.equal(out['image'], c['image']) and tuple(out['image'].stride()) == c['image_stride'], key
        assert torch.equal(out['label'], c['label']), key
        if c['norm']:
            assert float(out['label'].mean(dim=-1).abs().max()) < 1e-5


def test_mae_entry_script_and_loop_are_the_reference_names():
    """tools.train_mae_self_supervised_model (reference tools/train_mae_self_supervised_model.py) and
    tools.scripts.train_mae_self_supervised_learning (reference tools/scripts.py:1774) under the reference's spelling."""
    import importlib
    import inspect
    m = importlib.import_module('simpleaicv_pytorch_training_examples_amd.tools.train_mae_self_supervised_model')
    assert callable(m.main) and callable(m.parse_args)
    from simpleaicv_pytorch_training_examples_amd.tools import scripts
    assert list(inspect.signature(scripts.train_mae_self_supervised_learning).parameters) == [
        'train_loader', 'model', 'criterion', 'optimizer', 'scheduler', 'epoch', 'logger', 'config']
    import tools.scripts as alias                      # the reference spelling resolves to the same module object
    assert alias.train_mae_self_supervised_learning is scripts.train_mae_self_supervised_learning


# ------------------------------------------------------------------------------------------------ RandomErasing (host call)
def test_random_erasing_host_call_reproduces_the_reference_bit_for_bit():
    """RandomErasing.__call__ (reference classification/common.py:561-640) under the same numpy seed: the same boxes and the same
    fill values in all three modes, one or several boxes (tests/golden/random_erasing.pt, produced by the reference class);
    plan() must name exactly the erased rectangle(s) without consuming the per-pixel fill draws."""
    import numpy as np
    from conftest import load_golden
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import RandomErasing
    cases = load_golden('random_erasing')['cases']
    assert len(cases) == 36 and sum(c['changed'] > 0 for c in cases) >= 15
    for c in cases:
        np.random.seed(c['seed'])
        image = np.random.standard_normal((40, 48, 3)).astype(np.float32)
        before = image.copy()
        out = RandomErasing(**c['kwargs'])({'image': image, 'label': 3})
        assert out['label'] == 3 and torch.equal(torch.from_numpy(out['image']), c['image']), (c['kwargs'], c['seed'])
        # the plan: same first draws -> same first box; its rectangle is where the reference changed pixels (single-box cases)
        np.random.seed(c['seed'])
        np.random.standard_normal((40, 48, 3))
        plan = RandomErasing(**c['kwargs']).plan(40, 48, 3)
        changed = (c['image'].numpy() != before).any(axis=-1)
        if 'max_count' not in c['kwargs']:
            assert len(plan) == (1 if c['changed'] else 0)
            if plan:
                top, left, h, w, value = plan[0]
                mask = np.zeros((40, 48), dtype=bool)
                mask[top:top + h, left:left + w] = True
                assert (changed <= mask).all() and changed.sum() >= 0.97 * mask.sum()       # (a drawn value may equal the old one)
                assert (value is None) == (c['kwargs']['mode'] == 'pixel')


@pytest.mark.parametrize('case', ['van', 'convformer', 'dinov3convnext'])
def test_detection_van_convformer_trees_and_init_draws_are_the_references(case):
    """state_dict keys in registration order and every initial tensor (checksums of the reference's own construction under the same
    seed; oracle/make_golden_r04.py det_van_convformer) -- reference detection/models/backbones/van.py:52-105, convformer.py:43-97.
    Host-only: construction runs no kernel."""
    import os
    import torch
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.backbones import convformer, dinov3convnext, van
    fx = load_golden('det_van_convformer')['cases'][case]
    torch.manual_seed(0)
    m = {'van': van.VANBackbone, 'convformer': convformer.MetaFormerBackbone,
         'dinov3convnext': dinov3convnext.Dinov3ConvNeXtBackbone}[case](**fx['kwargs'])
    g = torch.Generator().manual_seed(33)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'layer_scale' in n or n.endswith('.scale') or n.endswith('.gamma'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2 + 0.5)
            elif n.endswith('.bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    sd = m.state_dict()
    assert list(sd.keys()) == list(fx['param_sum'].keys())
    for k, v in sd.items():
        assert abs(float(v.double().sum()) - fx['param_sum'][k]) <= 1e-6 * max(1.0, fx['param_abs_sum'][k]), k
    assert m.out_channels == fx['kwargs']['embedding_planes']


def _augment_image(seed, h=48, w=64):
    """the generator's test card (oracle/make_golden_r04.py _augment_image), restated"""
    import numpy as np
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([xx * 255 // (w - 1), yy * 255 // (h - 1), (xx + yy) * 255 // (h + w - 2)], axis=-1).astype(np.int64)
    img[h // 4:h // 2, w // 3:w // 2] = (240, 30, 120)
    return np.clip(img + rng.randint(-20, 21, size=img.shape), 0, 255).astype(np.uint8)


def test_auto_and_rand_augment_reproduce_the_reference_images():
    """AugmentOp for every op name at two magnitudes, AutoAugment's four policies and RandAugment in four configurations: the output
    bytes under the same `random` / numpy seeds equal what the reference produced (reference classification/auto_rand_augment.py:
    314-355, 538-565, 646-691; fixture oracle/make_golden_r04.py auto_rand_augment) -- so ops, magnitude maps, policy tables AND the
    order of the draws are the reference's."""
    import hashlib
    import json
    import os
    import random
    import numpy as np
    import PIL
    from PIL import Image
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import auto_rand_augment as ara
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import AutoAugment, Opencv2PIL, PIL2Opencv, RandAugment
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'auto_rand_augment.json')) as f:
        fx = json.load(f)
    if fx['pil_version'].split('.')[:1] != PIL.__version__.split('.')[:1]:
        pytest.skip(f'fixture made with Pillow {fx["pil_version"]}, this is {PIL.__version__}: resampling kernels may differ')

    def same(img, case):
        a = np.asarray(img)
        assert list(a.shape) == case['shape']
        assert hashlib.sha256(a.tobytes()).hexdigest() == case['sha256'], (case, float(a.mean()))

    assert sorted(ara.NAME_TO_OP) == sorted({c['name'] for c in fx['ops']}) and len(ara.NAME_TO_OP) == 24
    hp = dict(translate_const=28, img_mean=(124, 116, 104), magnitude_std=0.5)
    for c in fx['ops']:
        random.seed(c['seed'])
        same(ara.AugmentOp(c['name'], prob=1.0, magnitude=c['magnitude'], hparams=hp)(Image.fromarray(_augment_image(c['image_seed']))), c)
    augs = {p: AutoAugment(p, resize=64, magnitude_std=0.5 if p.endswith('r') else None) for p in ('original', 'originalr', 'v0', 'v0r')}
    assert all(len(a.policy) == 25 and all(len(sp) == 2 for sp in a.policy) for a in augs.values())
    for c in fx['auto']:
        random.seed(c['seed'])
        sample = Opencv2PIL()({'image': _augment_image(c['image_seed']).astype(np.float32), 'label': 3})
        out = augs[c['policy']](sample)
        assert out['label'] == 3
        same(out['image'], c)
    for c in fx['rand']:
        kw = {k: (float('inf') if v == 'inf' else v) for k, v in c['kwargs'].items()}
        random.seed(c['seed'])
        np.random.seed(c['np_seed'])
        out = RandAugment(resize=64, **kw)({'image': Image.fromarray(_augment_image(c['image_seed'])), 'label': 1})
        same(out['image'], c)
    back = PIL2Opencv()({'image': Image.fromarray(_augment_image(0)), 'label': 0})['image']
    assert back.dtype == np.float32 and np.array_equal(back, _augment_image(0).astype(np.float32))


@pytest.mark.parametrize('case', ['retinanet', 'fcos'])
def test_dinov3_vit_detector_trees_and_init_draws_are_the_references(case):
    """dinov3_vit_retinanet.RetinaNet / dinov3_vit_fcos.FCOS: state_dict keys in registration order and every initial tensor by
    checksum against the reference's own construction under the same seed (fixture dinov3_detectors); all twelve factories resolve
    their trunk by the reference's names (detection/models/dinov3_vit_retinanet.py:120-156, dinov3_vit_fcos.py:109-144)."""
    import torch
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection import models
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import backbones, dinov3_vit_fcos, dinov3_vit_retinanet
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.backbones.dinov3vit import DinoVisionTransformer, VitPyramidNeck
    gold = load_golden('dinov3_detectors')
    backbones.__dict__['tiny_dinov3_backbone'] = lambda pretrained_path='', **kw: DinoVisionTransformer(**gold['trunk'], **kw)
    torch.manual_seed(0)
    m = (dinov3_vit_retinanet.RetinaNet if case == 'retinanet' else dinov3_vit_fcos.FCOS)('tiny_dinov3_backbone', planes=64, num_classes=6)
    assert isinstance(m.neck, VitPyramidNeck)
    g = torch.Generator().manual_seed(44)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('.gamma'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3 + 1.0)
            elif n.endswith('.bias') and 'cls_out' not in n and 'cls_head' not in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    fx = gold['cases'][case]
    sd = m.state_dict()
    assert list(sd.keys()) == list(fx['param_sum'].keys())
    for k, v in sd.items():
        assert abs(float(v.double().sum()) - fx['param_sum'][k]) <= 1e-6 * max(1.0, fx['param_abs_sum'][k]), k
    for trunk in ('small', 'small_plus', 'base', 'large', 'large_plus', 'huge_plus'):
        for head in ('retinanet', 'fcos'):
            assert callable(models.__dict__[f'dinov3_vit_{trunk}_patch16_{head}'])
            assert f'dinov3_vit_{trunk}_patch16_backbone' in backbones.__dict__


def test_test_entry_scripts_and_configs_follow_the_reference_surface():
    """tools.test_classification_model / tools.test_detection_model (reference tools/test_classification_model.py:31-103,
    test_detection_model.py:29-98) exist under the reference's names with main / parse_args, tools.utils exposes
    compute_macs_and_params(config, model), and the two benchmark test_config.py files import and collate."""
    import importlib
    import importlib.util
    import inspect
    import os
    from conftest import ROOT
    for name in ('test_classification_model', 'test_detection_model'):
        m = importlib.import_module(f'simpleaicv_pytorch_training_examples_amd.tools.{name}')
        assert callable(m.main) and callable(m.parse_args)
        assert importlib.import_module(f'tools.{name}') is m
    from simpleaicv_pytorch_training_examples_amd.tools import utils
    assert list(inspect.signature(utils.compute_macs_and_params).parameters) == ['config', 'model']
    assert utils._with_unit(4089184256, 'MACs') == '4.089 GMACs' and utils._with_unit(25557032, '') == '25.557 M'
    assert utils._with_unit(8.2e9, 'FLOPS') == '8.2 GFLOPS' and utils._with_unit(512, 'MACs') == '512 MACs'
    os.environ.update(SAICV_CLS_TEST='8', SAICV_DET_TEST='4', SAICV_DET_TRAIN='4')
    try:
        for d, keys in (('00.classification_training/imagenet/resnet50', {'image': (2, 3, 224, 224), 'label': (2,)}),
                        ('03.detection_training/coco/res50_retinanet_yoloresize1024', {'image': (2, 3, 1024, 1024), 'annots': (2, 100, 5)})):
            spec = importlib.util.spec_from_file_location('test_cfg_' + d.split('/')[-1], os.path.join(ROOT, d, 'test_config.py'))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            c = mod.config
            assert c.batch_size in (256, 32) and c.seed == 0 and hasattr(c, 'test_criterion')
            batch = c.test_collater([c.test_dataset[i] for i in range(2)])
            for k, shp in keys.items():
                assert tuple(batch[k].shape) == shp, (d, k)
        assert c.eval_type == 'COCO' and c.decoder is not None
    finally:
        for k in ('SAICV_CLS_TEST', 'SAICV_DET_TEST', 'SAICV_DET_TRAIN'):
            os.environ.pop(k, None)


def test_every_environment_switch_is_documented_and_every_documented_switch_exists():
    """INTEGRATION.md's switch table is the spec of what runs by default (hygiene): every SAICV_* name read through
    getenv / os.environ anywhere in the product (csrc, package, bench.py, benchmark configs) has a row there, and no row names a
    switch the code no longer reads."""
    import glob
    import os
    import re
    from conftest import ROOT
    pkg = os.path.join(ROOT, 'simpleaicv_pytorch_training_examples_amd')
    files = [os.path.join(ROOT, 'bench.py')] + glob.glob(os.path.join(pkg, 'csrc', '*.hip')) + glob.glob(os.path.join(pkg, 'csrc', '*.h'))
    files += glob.glob(os.path.join(pkg, '**', '*.py'), recursive=True) + glob.glob(os.path.join(ROOT, '[0-9][0-9].*', '**', '*_config.py'), recursive=True)
    read = set()
    for f in files:
        text = open(f, errors='replace').read()
        read |= set(re.findall(r'getenv\(\s*"(SAICV_[A-Z0-9_]+)"', text))
        read |= set(re.findall(r"""environ(?:\.get\(|\[|\.pop\(|\.setdefault\()\s*['"](SAICV_[A-Z0-9_]+)['"]""", text))
        read |= set(re.findall(r'env_(?:int|flag|float|str)\(\s*"(SAICV_[A-Z0-9_]+)"', text))
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    documented = set(re.findall(r'SAICV_[A-Z0-9_]+', doc))
    prefixes = {m[:-1] for m in re.findall(r'SAICV_[A-Z0-9_]+_\*', doc)}           # families written as SAICV_DET_*
    missing = sorted(n for n in read if n not in documented and not any(n.startswith(p) for p in prefixes))
    assert not missing, f'switches read by the code without a row in INTEGRATION.md: {missing}'
    assert len(read) > 40
    rows = set(re.findall(r'^\| `(SAICV_[A-Z0-9_]+)`', doc, flags=re.M))
    stale = sorted(rows - read)
    assert not stale, f'INTEGRATION.md rows for switches nothing reads: {stale}'


def test_detr_pad_mask_from_sizes_equals_the_collaters_mask():
    """pad_mask_on_device(scaled_size, S, device) == the mask DETRDetectionCollater fills on the host (reference
    detection/common.py:315-322) for ragged image sizes, including a full-canvas image and a one-pixel one; the detection loop takes it
    when config.device_pad_mask is set."""
    import numpy as np
    import torch
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.common import DETRDetectionCollater, pad_mask_on_device
    rng = np.random.default_rng(0)
    sizes = [(64, 64), (1, 1), (37, 64), (64, 5), (20, 33)]
    data = [{'image': rng.random((h, w, 3), dtype=np.float32), 'annots': np.zeros((0, 5), dtype=np.float32),
             'scale': np.float32(1.0), 'size': np.array([h, w], dtype=np.float32)} for h, w in sizes]
    batch = DETRDetectionCollater(resize=64, resize_type='yolo_style', max_annots_num=4)(data)
    got = pad_mask_on_device(batch['scaled_size'], 64, torch.device('cpu'))
    assert got.dtype == torch.bool and got.shape == batch['mask'].shape
    assert torch.equal(got, batch['mask'])
    assert int((~got[1]).sum()) == 1 and not bool(got[0].any())
    import inspect
    from simpleaicv_pytorch_training_examples_amd.tools import scripts
    assert 'device_pad_mask' in inspect.getsource(scripts.train_detection)


def test_wgrad_carried_offsets_equal_rebuilt_offsets():
    """Host emulation of the offset recurrence of igemm_tn_dma_kernel<..., INCR> (csrc/igemm.hip, SAICV_TN_INCR=1): walking a DMA row
    32 output pixels at a time, offset += inc_base + carry1 * inc_cy1 + carry2 * inc_cy2 (mod 2^32) equals the offset rebuilt from
    (img, oh, ow) at every step, for random geometries / strides / paddings / taps -- the kernel itself has not run on a GPU yet."""
    import random
    rnd = random.Random(7)
    mask = 0xffffffff
    checked = 0
    for _ in range(1500):
        n, h, w = rnd.randint(1, 9), rnd.randint(3, 40), rnd.randint(3, 40)
        c, k, stride, pad = rnd.choice([8, 16, 64]), rnd.choice([1, 3, 7]), rnd.choice([1, 2, 4]), rnd.choice([0, 1, 3])
        oh_n, ow_n = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        if oh_n < 1 or ow_n < 1:
            continue
        br = 32
        d_img = br // (oh_n * ow_n)
        d_oh, d_ow = divmod(br - d_img * oh_n * ow_n, ow_n)
        m_total = n * oh_n * ow_n
        m, tap, c0 = rnd.randrange(m_total), rnd.randrange(k * k), rnd.randrange(0, c, 8)
        fr, fs = tap // k - pad, tap % k - pad
        img, rem = divmod(m, oh_n * ow_n)
        oh, ow = divmod(rem, ow_n)
        inc_base = ((((d_img * h + stride * d_oh) * w + stride * d_ow) * c) * 2) & mask
        inc_cy1 = (((stride * w - stride * ow_n) * c) * 2) & mask
        inc_cy2 = ((((h - stride * oh_n) * w) * c) * 2) & mask
        off = ((((img * h + oh * stride + fr) * w + ow * stride + fs) * c + c0) * 2) & mask
        step = 0
        while m + step * br < m_total:
            rebuilt = ((((img * h + oh * stride + fr) * w + ow * stride + fs) * c + c0) * 2) & mask
            assert rebuilt == off, (n, h, w, c, k, stride, pad, m, tap, step)
            ow += d_ow
            c1 = ow >= ow_n
            ow -= ow_n if c1 else 0
            oh += d_oh + (1 if c1 else 0)
            c2 = oh >= oh_n
            oh -= oh_n if c2 else 0
            img += d_img + (1 if c2 else 0)
            off = (off + inc_base + (inc_cy1 if c1 else 0) + (inc_cy2 if c2 else 0)) & mask
            step += 1
            checked += 1
    assert checked > 5000
