"""Host logic of the loop library (Scheduler, optimizer parameter grouping) against fixtures
produced by the reference's own tools/utils.py (oracle/make_host_golden.py)."""
import json
import os

import pytest
import torch

from conftest import GOLDEN
from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones
from simpleaicv_pytorch_training_examples_amd.tools import utils as U

FX = json.load(open(os.path.join(GOLDEN, 'host_logic.json')))


class Cfg:
    pass


class FakeOptimizer:
    def __init__(self, groups):
        self.param_groups = groups


def _cfg(case):
    cfg = Cfg()
    cfg.optimizer = tuple(case['optimizer'])
    cfg.scheduler = tuple(case['scheduler'])
    cfg.epochs = case['epochs']
    return cfg


def _groups(cfg, model):
    """what build_optimizer hands to the fused optimizer, without constructing it (needs a GPU)"""
    groups, index = [], {}
    for name, p, wd, lr, scale in U._per_parameter_settings(cfg, model):
        key = (wd, lr, scale)
        if key not in index:
            index[key] = len(groups)
            groups.append({'params': [], 'names': [], 'weight_decay': wd, 'lr': lr * (scale if scale is not None else 1.)})
        groups[index[key]]['params'].append(p)
        groups[index[key]]['names'].append(name)
    return groups


@pytest.mark.parametrize('key', sorted(FX.keys()))
def test_grouping_and_schedule_match_reference(key):
    case = FX[key]
    if case['network'] not in backbones.__dict__:
        pytest.skip(f"{case['network']} not built yet")
    torch.manual_seed(0)
    model = backbones.__dict__[case['network']](**case['kwargs'])
    cfg = _cfg(case)
    groups = _groups(cfg, model)
    eff = {n: [g['lr'], g['weight_decay']] for g in groups for n in g['names']}
    assert set(eff) == set(case['effective'])
    for n, (lr, wd) in case['effective'].items():
        assert eff[n][0] == pytest.approx(lr, rel=1e-12), n
        assert eff[n][1] == pytest.approx(wd, rel=1e-12), n
    assert len(groups) == len(case['groups'])
    opt = FakeOptimizer(groups)
    sched = U.Scheduler(cfg, opt)
    for point in case['schedule']:
        sched.step(opt, point['epoch'])
        assert sched.current_lr == pytest.approx(point['current_lr'], rel=1e-12, abs=1e-18)
        mine = {g['names'][0]: g['lr'] for g in groups}
        # the reference keys each group by its first parameter; compare per parameter instead
        per_param = {n: g['lr'] for g in groups for n in g['names']}
        for first_name, lr in point['group_lrs'].items():
            assert per_param[first_name] == pytest.approx(lr, rel=1e-12, abs=1e-18), (point['epoch'], first_name)


def test_scheduler_state_dict_roundtrip():
    case = FX['resnet50_sgd']
    cfg = _cfg(case)
    opt = FakeOptimizer([{'lr': 0.1}, {'lr': 0.1}])
    s = U.Scheduler(cfg, opt)
    s.step(opt, 31.5)
    sd = s.state_dict()
    s2 = U.Scheduler(cfg, FakeOptimizer([{'lr': 0.1}, {'lr': 0.1}]))
    s2.load_state_dict(sd)
    assert s2.current_lr == s.current_lr and s2.init_param_groups_lr == s.init_param_groups_lr


def test_collater_layout_and_meters():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import (AccMeter, AverageMeter,
                                                                                         ClassificationCollater)
    import numpy as np
    batch = [{'image': np.random.rand(8, 6, 3).astype(np.float32), 'label': i} for i in range(4)]
    out = ClassificationCollater()(batch)
    assert out['image'].shape == (4, 3, 8, 6) and out['image'].stride() == (144, 1, 18, 3)   # NHWC-strided view
    assert out['label'].dtype == torch.int64
    m = AverageMeter()
    m.update(2.0, 3)
    m.update(4.0, 1)
    assert m.avg == pytest.approx(2.5)
    a = AccMeter()
    a.update(3, 4, 8)
    a.compute()
    assert a.acc1 == 0.375 and a.acc5 == 0.5


def test_sam_collater_contract():
    """SAMBatchCollater (reference interactive_segmentation/common.py:129-232): square zero-padded canvas at
    the top-left, TRUE NCHW image batch, mask / prompt-mask with a channel axis, prompt mask resized with
    OpenCV's nearest rule floor(dst * src / dst_size) onto a (resize / 4) canvas."""
    import numpy as np
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.common import (SAMBatchCollater,
                                                                                                   resize_nearest)
    rng = np.random.RandomState(0)
    samples = []
    for h, w in ((128, 96), (64, 128)):
        mask = (rng.rand(h, w) > 0.5).astype(np.float32)
        samples.append({'image': rng.randn(h, w, 3).astype(np.float32), 'box': np.array([1, 2, 30, 40], dtype=np.float32),
                        'mask': mask, 'size': np.array([h, w], dtype=np.float32),
                        'prompt_point': np.array([[5, 6, 1]], dtype=np.float32),
                        'prompt_box': np.array([1, 2, 30, 40], dtype=np.float32), 'prompt_mask': mask.copy()})
    out = SAMBatchCollater(resize=128)(samples)
    assert out['image'].shape == (2, 3, 128, 128) and out['image'].dtype == torch.float32 and out['image'].is_contiguous()
    assert torch.equal(out['image'][0, :, :128, :96], torch.from_numpy(samples[0]['image']).permute(2, 0, 1))
    assert float(out['image'][0, :, :, 96:].abs().sum()) == 0 and float(out['image'][1, :, 64:, :].abs().sum()) == 0
    assert out['mask'].shape == (2, 1, 128, 128) and out['prompt_mask'].shape == (2, 1, 32, 32)
    assert out['prompt_point'].shape == (2, 1, 3) and out['prompt_box'].shape == (2, 4) and out['size'].shape == (2, 2)
    # sample 0: 128x96 -> factor 0.25 -> 32x24, index rule floor(i * 4)
    assert torch.equal(out['prompt_mask'][0, 0, :32, :24], torch.from_numpy(samples[0]['mask'][::4, ::4]))
    assert float(out['prompt_mask'][0, 0, :, 24:].sum()) == 0
    a = np.arange(35, dtype=np.float32).reshape(5, 7)
    r = resize_nearest(a, 3, 2)                       # cols floor(x * 7/3) = 0, 2, 4 ; rows floor(y * 5/2) = 0, 2
    assert np.array_equal(r, a[[0, 2]][:, [0, 2, 4]])


def test_mixup_cutmix_collater_matches_reference_fixture():
    """Same numpy seed -> same mixing plan, images and soft labels as the reference's collater
    (oracle/make_golden_mixup.py ran reference mixupcutmixclassificationcollator.py:99-284)."""
    import numpy as np
    from conftest import load_golden
    from oracle.make_golden_mixup import batch
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import MixupCutmixClassificationCollater
    for case in load_golden('mixup_cutmix'):
        np.random.seed(case['np_seed'])
        got = MixupCutmixClassificationCollater(num_classes=10, **case['kwargs'])(batch(case['data_seed']))
        assert got['image'].shape == case['image'].shape and got['label'].shape == case['label'].shape
        assert float((got['image'] - case['image']).abs().max()) < 1e-6, case['kwargs']
        assert float((got['label'] - case['label']).abs().max()) < 1e-6, case['kwargs']
        assert got['image'].stride()[1] == 1          # NHWC-strided view, as the reference's permute returns it


def test_benchmark_configs_import_like_reference_configs_and_collate():
    """The seven benchmark copies of the reference train_config.py files import through the `SimpleAICV` /
    `tools` aliases exactly as the reference spells them, build their model on the CPU, and their dataset + collater
    produce the loader contract of each loop."""
    import importlib.util
    import os
    from conftest import ROOT
    expect = {
        '00.classification_training/cifar100/resnet18cifar': ('resnet18cifar', {'image': (2, 3, 32, 32), 'label': (2,)}),
        '00.classification_training/imagenet/resnet50': ('resnet50', {'image': (2, 3, 224, 224), 'label': (2,)}),
        '00.classification_training/imagenet/vit_base_patch16_for_self_train_mae_pretrain':
            ('vit_base_patch16', {'image': (2, 3, 224, 224), 'label': (2, 1000)}),
        '02.masked_image_modeling_training/imagenet/mae_vit_base_patch16_224':
            ('vit_base_patch16_224_mae_pretrain_model', {'image': (2, 3, 224, 224), 'label': (2, 196, 768)}),
        '03.detection_training/coco/res50_detr_yoloresize1024':
            ('resnet50_detr', {'image': (2, 3, 1024, 1024), 'mask': (2, 1024, 1024), 'scaled_annots': (2, 100, 5)}),
        '03.detection_training/coco/res50_retinanet_yoloresize1024':
            ('resnet50_retinanet', {'image': (2, 3, 1024, 1024), 'annots': (2, 100, 5)}),
        '03.detection_training/coco/res50_fcos_yoloresize1024':
            ('resnet50_fcos', {'image': (2, 3, 1024, 1024), 'annots': (2, 100, 5)}),
        '13.interactive_segmentation_training/13.1.sam_segmentation_training/sam_b_training':
            ('sam_b', {'image': (2, 3, 1024, 1024), 'mask': (2, 1, 1024, 1024), 'prompt_point': (2, 1, 3), 'prompt_box': (2, 4),
                       'prompt_mask': (2, 1, 256, 256)}),
    }
    for d, (network, shapes) in expect.items():
        spec = importlib.util.spec_from_file_location('cfg_' + network, os.path.join(ROOT, d, 'train_config.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        c = mod.config
        assert c.network == network and c.optimizer[0] in ('SGD', 'AdamW') and c.use_amp is True
        batch = c.train_collater([c.train_dataset[i] for i in range(2)])
        for k, shp in shapes.items():
            assert tuple(batch[k].shape) == shp, (d, k, tuple(batch[k].shape))
    import SimpleAICV.classification.backbones as a
    import simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones as b
    assert a is b                                     # one module object under both spellings


def test_ema_average_does_not_move_on_a_skipped_iteration():
    """EmaModel.update(model, skip_flag): the device-side skip flag gates the average like the reference's `continue`
    (tools/scripts.py:196-200) without a host read."""
    import torch
    from simpleaicv_pytorch_training_examples_amd.tools.utils import EmaModel
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    ema = EmaModel(m, decay=0.9)
    with torch.no_grad():
        m[0].weight.add_(1.0)
        m[1].num_batches_tracked.add_(5)
    w0 = ema.ema_model[0].weight.clone()
    ema.update(m, torch.tensor([1.0]))
    assert torch.equal(ema.ema_model[0].weight, w0)
    ema.update(m, torch.tensor([0.0]))
    assert torch.allclose(ema.ema_model[0].weight, w0 + 0.1 * (m[0].weight - w0))
    ema.update(m)
    assert int(ema.ema_model[1].num_batches_tracked) == 5


def test_zero_pool_hands_out_zeroed_distinct_slices_and_resets_at_the_step_boundary():
    """ops._ZeroPool (the scratch BatchNorm statistics are added into with atomics): slices never overlap within a step,
    come back zeroed after the step boundary, keep their addresses from step to step, and the pool stays bounded when
    nobody announces a step boundary."""
    import torch
    from simpleaicv_pytorch_training_examples_amd import ops
    pool = ops._ZeroPool
    saved = (pool.chunks, pool.handed, pool.LIMIT)
    pool.chunks, pool.handed = [], 0
    try:
        dev = torch.device('cpu')
        a = pool.take(2 * 8 * 40, dev)
        b = pool.take(100, dev)
        assert a.numel() == 640 and b.numel() == 100 and float(a.abs().sum()) == 0 and float(b.abs().sum()) == 0
        assert a.data_ptr() % 256 == b.data_ptr() % 256                     # slices start on 256-byte boundaries
        a.fill_(1.0)
        b.fill_(2.0)
        assert float(a.sum()) == 640 and float(b.sum()) == 200              # distinct memory
        ptr_a = a.data_ptr()
        ops.bump_weights_epoch()                                            # the step boundary
        a2 = pool.take(2 * 8 * 40, dev)
        assert a2.data_ptr() == ptr_a and float(a2.abs().sum()) == 0 and float(a.abs().sum()) == 0
        pool.LIMIT = 1000
        c = pool.take(5000, dev)                                            # beyond the limit without a boundary:
        d = pool.take(64, dev)                                              # plain zero tensors, the pool does not grow
        assert float(c.abs().sum()) == 0 and float(d.abs().sum()) == 0 and sum(ch[0].numel() for ch in pool.chunks) <= pool.CHUNK + 5056
        assert [ops._stat_rows(t) for t in (1, 7, 8, 63, 64, 511, 512, 3136)] == [1, 1, 2, 2, 4, 4, 8, 8]
    finally:
        pool.chunks, pool.handed, pool.LIMIT = saved


def test_load_state_dict_filters_and_resizes_the_position_embedding_like_the_reference(tmp_path):
    """classification/common.py load_state_dict (reference :758-840) against what the reference itself produced
    (oracle/make_golden_loadstate.py): name / shape / excluded-layer filtering, bicubic resize of a 4 x 4 position grid to
    6 x 6 with the class-token row kept, buffers included."""
    import torch
    from oracle.make_golden_loadstate import Toy
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import load_state_dict
    fx = torch.load(os.path.join(GOLDEN, 'load_state_dict.pt'), weights_only=False)
    path = str(tmp_path / 'saved.pth')
    torch.save(fx['saved'], path)
    for key, case in fx['results'].items():
        torch.manual_seed(fx['model_seed'])
        model = Toy(6)
        before = {k: v.clone() for k, v in model.state_dict().items()}
        load_state_dict(path, model, **case['kwargs'])
        after = model.state_dict()
        assert sorted(k for k in after if not torch.equal(after[k], before[k])) == case['changed'], key
        for k, v in case['after'].items():
            assert torch.allclose(after[k].double(), v.double(), rtol=0, atol=1e-6), (key, k)


def test_bench_spawn_gives_every_rank_the_torchrun_environment(monkeypatch):
    """`python bench.py --gpus N` (bench.spawn): N worker processes with RANK / LOCAL_RANK / WORLD_SIZE, a 127.0.0.1
    rendezvous on one free port, dmabuf IPC mode for RCCL; only rank 0 keeps stdout (the ONE JSON line); a failing rank fails
    the run and takes the others down."""
    import importlib
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            self.env, self.stdout, self.rank = env, stdout, int(env['RANK'])
            self.terminated = False
            started.append(self)

        def poll(self):
            return 3 if self.rank == 1 else (-15 if self.terminated else None)

        def terminate(self):
            self.terminated = True

        def kill(self):
            self.terminated = True

    monkeypatch.setattr(bench.subprocess, 'Popen', FakeProc)
    monkeypatch.delenv('HSA_ENABLE_IPC_MODE_LEGACY', raising=False)
    args = type('A', (), {'gpus': 4})()
    rc = bench.spawn(args)
    assert rc == 3 and len(started) == 4
    ports = {p.env['MASTER_PORT'] for p in started}
    assert len(ports) == 1
    for r, p in enumerate(started):
        assert (p.env['RANK'], p.env['LOCAL_RANK'], p.env['WORLD_SIZE'], p.env['MASTER_ADDR']) == (str(r), str(r), '4', '127.0.0.1')
        assert p.env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and p.env['SAICV_BENCH_SPAWNED'] == '1'
        assert (p.stdout is None) == (r == 0)
    assert all(p.terminated for p in started if p.rank not in (1,))


def test_bench_runs_the_captured_overlapped_step_by_default_with_several_ranks():
    """the path `bench.py --gpus N` takes must be the one whose gradient all-reduces overlap backward --
    the captured step (collectives on the communication stream behind graph edges), not eager launches with the collectives
    serialised on the compute stream.  bench.want_step_graph is the selection; eager stays reachable explicitly."""
    import importlib
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    for world in (2, 4, 8):
        assert bench.want_step_graph(False, False, world, None) is True          # the driver's launch: no flags, no env
        assert bench.want_step_graph(False, True, world, None) is True
        assert bench.want_step_graph(True, False, world, None) is False          # --eager
        assert bench.want_step_graph(False, False, world, '0') is False          # SAICV_STEP_GRAPH=0
    assert bench.want_step_graph(False, False, 1, None) is True
    assert bench.want_step_graph(True, False, 1, None) is False
    src = open(bench.__file__).read()
    assert "'overlap': bool(use_graph) if world > 1 else None" in src            # ... and the JSON line says which one ran


def test_anchor_and_position_tables_are_bit_identical_to_the_reference():
    """models/anchor.py against digests of the reference's tables (oracle/make_golden_fcos.py): the default five-level pyramid on
    a square and a ragged image, and a custom three-level configuration"""
    import hashlib
    import os
    import numpy as np
    import torch
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models.anchor import RetinaAnchors, FCOSPositions
    fx = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'anchors_positions.pt'), weights_only=False)

    def same(a, d):
        a = np.ascontiguousarray(a)
        assert tuple(a.shape) == tuple(d['shape']) and str(a.dtype) == d['dtype']
        assert np.array_equal(a.reshape(-1)[:8], d['first'].numpy()) and np.array_equal(a.reshape(-1)[-8:], d['last'].numpy())
        assert hashlib.sha256(a.tobytes()).hexdigest() == d['sha256']

    for key, t in fx.items():
        if key == 'custom':
            gen = RetinaAnchors(areas=[[24, 24], [48, 48], [96, 96]], ratios=[0.4, 1.6], scales=[1.0, 1.5], strides=[8, 16, 32])
            for a, d in zip(gen(t['sizes']), t['anchors']):
                same(a, d)
            continue
        for a, d in zip(RetinaAnchors()(t['sizes']), t['anchors']):
            same(a, d)
        for a, d in zip(FCOSPositions()(t['sizes']), t['positions']):
            same(a, d)


def test_dense_decoder_host_stages_match_the_reference():
    """DecodeMethod / DetNMSMethod / box decoding of SimpleAICV/detection/decode.py on the host, fed with ALL anchors the way the
    reference feeds them (arg-max in numpy here; on the GPU the decoders take it from csrc/detloss.hip -- tests/test_gpu_decoders.py),
    against the detections the reference decoders produced (oracle/make_golden_decoders.py): identical scores, classes and boxes."""
    import os
    import sys
    import numpy as np
    import torch
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.decode import RetinaDecoder, FCOSDecoder
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'oracle'))
    try:
        import make_golden_decoders as m
    finally:
        sys.path.pop(0)
    fx = torch.load(os.path.join(root, 'tests', 'golden', 'dense_decoders.pt'), weights_only=True)
    cls, reg = m.retina_inputs()
    b = cls[0].shape[0]
    probs = np.concatenate([t.numpy().reshape(b, -1, t.shape[-1]) for t in cls], axis=1)
    regs = np.concatenate([t.numpy().reshape(b, -1, 4) for t in reg], axis=1)
    classes = probs.argmax(axis=2)
    scores = np.take_along_axis(probs, classes[..., None], axis=2)[..., 0]
    for name, ref in fx['retina'].items():
        dec = RetinaDecoder(**ref['config'])
        table = np.concatenate([a.reshape(-1, 4) for a in dec.anchors([[t.shape[2], t.shape[1]] for t in cls])], axis=0)
        boxes = dec.snap_txtytwth_to_x1y1x2y2(regs, np.repeat(table[None], b, axis=0))
        s, c, bx = dec.decode_function(scores, classes, boxes)
        assert np.array_equal(s, ref['scores'].numpy()) and np.array_equal(c, ref['classes'].numpy()) and np.array_equal(bx, ref['boxes'].numpy()), name
    cls, reg, ctr = m.fcos_inputs()
    probs = np.concatenate([t.numpy().reshape(b, -1, t.shape[-1]) for t in cls], axis=1)
    regs = np.concatenate([t.numpy().reshape(b, -1, 4) for t in reg], axis=1)
    ctrs = np.concatenate([t.numpy().reshape(b, -1) for t in ctr], axis=1)
    classes = probs.argmax(axis=2)
    scores = np.sqrt(np.take_along_axis(probs, classes[..., None], axis=2)[..., 0] * ctrs)
    for name, ref in fx['fcos'].items():
        dec = FCOSDecoder(**ref['config'])
        table = np.concatenate([p.reshape(-1, 2) for p in dec.positions([[t.shape[2], t.shape[1]] for t in cls])], axis=0)
        boxes = dec.snap_ltrb_to_x1y1x2y2(regs, np.repeat(table[None], b, axis=0))
        s, c, bx = dec.decode_function(scores, classes, boxes)
        assert np.array_equal(s, ref['scores'].numpy()) and np.array_equal(c, ref['classes'].numpy()) and np.array_equal(bx, ref['boxes'].numpy()), name


@pytest.mark.parametrize('variant', ['all_classes', 'class_without_ground_truth'])
def test_voc_detection_evaluation_matches_the_reference(variant):
    """tools.scripts.evaluate_voc_detection / test_detection against the result dict the REFERENCE produced from the same stub
    model / criterion / decoder (oracle/make_golden_voc_eval.py): loss, mAP at ten IoU thresholds, per-class AP -- and the NaN the
    reference reports when a class has detections but no ground truth."""
    import math
    import os
    import sys
    from simpleaicv_pytorch_training_examples_amd.tools import scripts
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'oracle'))
    try:
        import make_golden_voc_eval as m
    finally:
        sys.path.pop(0)
    fx = torch.load(os.path.join(root, 'tests', 'golden', 'voc_eval.pt'), weights_only=False)[variant]
    loader, model, criterion, decoder, config = m.stubs(variant == 'class_without_ground_truth')
    res = scripts.test_detection(loader, model, criterion, decoder, config)
    assert list(res.keys()) == list(fx.keys())
    assert abs(res['test_loss'] - fx['test_loss']) < 1e-6

    def same(a, b):
        return (math.isnan(a) and math.isnan(b)) or abs(a - b) <= 1e-9 * max(1.0, abs(b))

    for k, v in fx.items():
        if k.endswith('mAP'):
            assert same(float(res[k]), v), (k, res[k], v)
        elif k.endswith('per_class_ap'):
            assert set(res[k].keys()) == set(v.keys())
            for c, ap in v.items():
                assert same(float(res[k][c]), ap), (k, c, res[k][c], ap)


def test_coco_evaluation_runs_through_the_reference_result_keys():
    """tools.scripts.test_detection with eval_type = 'COCO' (reference tools/scripts.py:742-881): the stub model / criterion /
    decoder of the VOC fixture replayed through the numpy COCO protocol -- the reference's twelve result keys in its order, values
    in [0, 100], mAP@0.5 >= mAP@0.5:0.95, and the same detections scored against themselves give 100."""
    import os
    import sys
    from simpleaicv_pytorch_training_examples_amd.tools import cocoeval_numpy as CE
    from simpleaicv_pytorch_training_examples_amd.tools import scripts
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'oracle'))
    try:
        import make_golden_voc_eval as m
    finally:
        sys.path.pop(0)
    loader, model, criterion, decoder, config = m.stubs(False)
    config.eval_type = 'COCO'
    config.test_dataset = object()          # no .coco / .image_ids: ground truth comes from the loader's own annotations
    res = scripts.test_detection(loader, model, criterion, decoder, config)
    keys = list(res.keys())
    assert keys[:3] == ['test_loss', 'per_image_load_time', 'per_image_inference_time'] and keys[3:] == list(CE.STAT_NAMES)
    vals = [res[k] for k in CE.STAT_NAMES]
    assert all(v == -100 or 0.0 <= v <= 100.0 for v in vals), vals
    assert res[CE.STAT_NAMES[1]] >= res[CE.STAT_NAMES[0]] > 0.0
    assert res[CE.STAT_NAMES[8]] >= res[CE.STAT_NAMES[7]] >= res[CE.STAT_NAMES[6]]      # recall grows with the detection budget


@pytest.mark.parametrize('case', ['yolo', 'retina'])
def test_detection_collater_matches_the_reference(case):
    """DetectionCollater (RetinaNet / FCOS batches) against the batch the REFERENCE collater built from the same samples
    (oracle/make_golden_collater.py): canvas values and its NHWC-strided NCHW view, padded annotations, scale / size arrays"""
    import os
    import sys
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.common import DetectionCollater
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'oracle'))
    try:
        import make_golden_collater as m
    finally:
        sys.path.pop(0)
    fx = torch.load(os.path.join(root, 'tests', 'golden', 'detection_collater.pt'), weights_only=True)[case]
    out = DetectionCollater(**fx['config'])(m.samples())
    assert set(out) == {'image', 'annots', 'scale', 'size'}
    assert out['image'].dtype == torch.float32 and tuple(out['image'].stride()) == tuple(fx['image_stride'])
    assert torch.equal(out['image'], fx['image']) and torch.equal(out['annots'], fx['annots'])
    assert torch.equal(torch.from_numpy(out['scale']), fx['scale']) and torch.equal(torch.from_numpy(out['size']), fx['size'])
