"""The restated training loop and entry script on a real MI355X: loss goes down, a poisoned
batch is skipped on-device without touching parameters, checkpoints follow the reference schema
and training resumes from latest.pth (reference tools/train_classification_model.py:139-160,
:209-262).  The reference-trajectory and captured-vs-eager comparisons live in
tests/test_zz_gpu_trajectories.py (deterministic mode, collected last)."""
import logging
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


class SyntheticSet(torch.utils.data.Dataset):
    """HWC float images whose label is recoverable from the pixels (which quadrant is brighter /
    darker); 8 classes because the reference's eval loop takes top-5 (tools/scripts.py:73)."""

    def __init__(self, n=512, classes=8, size=32, seed=0, poison=()):
        g = torch.Generator().manual_seed(seed)
        self.labels = torch.randint(0, classes, (n,), generator=g)
        self.images = torch.randn(n, size, size, 3, generator=g) * 0.5
        h = size // 2
        for i, l in enumerate(self.labels.tolist()):
            r, c = divmod(l % 4, 2)
            self.images[i, r * h:(r + 1) * h, c * h:(c + 1) * h] += 1.5 if l < 4 else -1.5
        for i in poison:
            self.images[i, 0, 0, 0] = float('nan')

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        return {'image': self.images[i].numpy(), 'label': int(self.labels[i])}


def _config(dataset, batch=64, acc=1):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones, losses

    class config:
        pass
    torch.manual_seed(0)
    config.network = 'resnet18cifar'
    config.model = backbones.resnet18cifar(num_classes=8)
    config.train_criterion = losses.CELoss()
    config.train_dataset = dataset
    config.batch_size = batch
    config.accumulation_steps = acc
    config.optimizer = ('SGD', {'lr': 0.05, 'momentum': 0.9, 'global_weight_decay': False, 'weight_decay': 5e-4,
                                'no_weight_decay_layer_name_list': []})
    config.scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [60]})
    config.epochs = 3
    config.print_interval = 2
    config.use_amp = True
    config.use_ema_model = False
    config.local_rank = 0
    config.group = None
    config.gpus_num = 1
    config.sync_bn = False
    return config


def _loader(config):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import ClassificationCollater
    return torch.utils.data.DataLoader(config.train_dataset, batch_size=config.batch_size, shuffle=False,
                                       drop_last=True, collate_fn=ClassificationCollater())


def test_train_classification_learns_and_skips_poisoned_batches(caplog):
    from simpleaicv_pytorch_training_examples_amd.tools import scripts, utils
    config = _config(SyntheticSet(poison=(70,)))          # sample 70 sits in the 2nd batch
    model = config.model.cuda()
    optimizer, _ = utils.build_optimizer(config, model)
    scheduler = utils.Scheduler(config, optimizer)
    model, config.ema_model, config.scaler = utils.build_training_mode(config, model)
    logger = logging.getLogger('saicv_test')
    logger.setLevel(logging.INFO)
    loader = _loader(config)
    with caplog.at_level(logging.INFO, logger='saicv_test'):
        l1 = scripts.train_classification(loader, model, config.train_criterion, optimizer, scheduler, 1, logger, config)
        l2 = scripts.train_classification(loader, model, config.train_criterion, optimizer, scheduler, 2, logger, config)
        l3 = scripts.train_classification(loader, model, config.train_criterion, optimizer, scheduler, 3, logger, config)
    assert l3 < l1 * 0.6, (l1, l2, l3)
    assert caplog.text.count('skip this batch!') == 3          # once per epoch
    assert 'train: epoch 0001, iter [00004, 00008]' in caplog.text
    for p in model.parameters():
        assert torch.isfinite(p).all()


def test_poisoned_batch_leaves_parameters_untouched():
    from simpleaicv_pytorch_training_examples_amd.tools import scripts, utils
    ds = SyntheticSet(n=64, poison=(3,))
    config = _config(ds, batch=64)
    model = config.model.cuda()
    optimizer, _ = utils.build_optimizer(config, model)
    scheduler = utils.Scheduler(config, optimizer)
    model, config.ema_model, config.scaler = utils.build_training_mode(config, model)
    before = model.arena.flat_param.clone()
    logger = logging.getLogger('saicv_test2')
    scripts.train_classification(_loader(config), model, config.train_criterion, optimizer, scheduler, 1, logger, config)
    assert torch.equal(before, model.arena.flat_param)
    assert float(optimizer.momentum_buf.abs().sum()) == 0.0


def test_entry_script_checkpoints_and_resumes(tmp_path):
    work = tmp_path / 'work'
    work.mkdir()
    (work / 'train_config.py').write_text(textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {str(ROOT)!r})
        sys.path.insert(0, {str(os.path.join(ROOT, "tests"))!r})
        import torch
        from test_gpu_train_loop import SyntheticSet
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones, losses
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.common import ClassificationCollater

        class config:
            network = 'resnet18cifar'
            num_classes = 8
            input_image_size = 32
            model = backbones.__dict__[network](**{{'num_classes': num_classes}})
            train_criterion = losses.CELoss()
            test_criterion = losses.CELoss()
            train_dataset = SyntheticSet(n=256, seed=0)
            test_dataset = SyntheticSet(n=128, seed=1)
            train_collater = ClassificationCollater()
            test_collater = ClassificationCollater()
            seed = 0
            batch_size = 64
            num_workers = 0
            accumulation_steps = 2
            optimizer = ('SGD', {{'lr': 0.05, 'momentum': 0.9, 'global_weight_decay': False, 'weight_decay': 5e-4,
                                 'no_weight_decay_layer_name_list': []}})
            scheduler = ('MultiStepLR', {{'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [60]}})
            epochs = EPOCHS
            print_interval = 2
            sync_bn = False
            use_amp = True
            use_compile = False
            compile_params = {{}}
            use_ema_model = True
            ema_model_decay = 0.9
    '''))

    def run(epochs):
        cfg = (work / 'train_config.py').read_text().replace('EPOCHS', str(epochs))
        (work / 'train_config.py').write_text(cfg)
        env = dict(os.environ, PYTHONPATH=str(ROOT), MASTER_ADDR='127.0.0.1')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr',
               '127.0.0.1', '--master-port', '29533', '-m',
               'simpleaicv_pytorch_training_examples_amd.tools.train_classification_model', '--work-dir', str(work)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout + r.stderr

    out = run(2)
    ck = torch.load(work / 'checkpoints' / 'latest.pth', map_location='cpu', weights_only=True)
    assert set(ck) == {'epoch', 'time', 'best_acc1', 'test_loss', 'lr', 'model_state_dict', 'ema_model_state_dict',
                       'optimizer_state_dict', 'scheduler_state_dict'}
    assert ck['epoch'] == 2
    assert all(k.startswith('module.') for k in ck['model_state_dict'])
    assert 'module.conv1.layer.0.weight' in ck['model_state_dict']
    assert 'module.conv1.layer.1.running_mean' in ck['ema_model_state_dict']
    assert 'train done. model: resnet18cifar' in out
    finals = [f for f in os.listdir(work / 'checkpoints') if f.startswith('resnet18cifar-acc')]
    assert len(finals) == 1
    # resume: ask for one more epoch -> only epoch 3 runs
    (work / 'train_config.py').write_text((work / 'train_config.py').read_text().replace('epochs = 2', 'epochs = 3'))
    out2 = run(3)
    assert 'resuming model from' in out2 and 'resume_epoch: 002' in out2
    assert 'train: epoch 003' in out2 and 'train: epoch 001' not in out2
    ck2 = torch.load(work / 'checkpoints' / 'latest.pth', map_location='cpu', weights_only=True)
    assert ck2['epoch'] == 3


# ------------------------------------------------------------------------------------------ SAM loop
class SyntheticSamSet(torch.utils.data.Dataset):
    """Samples in the post-transform layout SAMBatchCollater consumes (interactive_segmentation/common.py)."""

    def __init__(self, n=8, size=256):
        self.n, self.size = n, size

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        import numpy as np
        rng = np.random.RandomState(i)
        s = self.size
        h, w = s, s - 32                                    # non-square: exercises the zero-padded canvas
        x1, y1 = rng.randint(8, w // 2), rng.randint(8, h // 2)
        bw, bh = rng.randint(24, w // 2 - 8), rng.randint(24, h // 2 - 8)
        mask = np.zeros((h, w), dtype=np.float32)
        mask[y1:y1 + bh, x1:x1 + bw] = 1.0
        image = rng.randn(h, w, 3).astype(np.float32) + mask[:, :, None]
        box = np.array([x1, y1, x1 + bw, y1 + bh], dtype=np.float32)
        return {'image': image, 'box': box.copy(), 'mask': mask, 'size': np.array([h, w], dtype=np.float32),
                'prompt_point': np.array([[x1 + bw // 2, y1 + bh // 2, 1.0]], dtype=np.float32),
                'prompt_box': box.copy(), 'prompt_mask': mask.copy()}


def test_train_sam_segmentation_runs_and_updates(caplog):
    import numpy as np
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation import losses
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.common import SAMBatchCollater
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything import sam
    from simpleaicv_pytorch_training_examples_amd.tools import interactive_segmentation_scripts as iss, utils

    class config:
        pass
    torch.manual_seed(0)
    np.random.seed(0)
    config.network = 'sam_tiny'
    config.input_image_size = 256
    config.model = sam.SAM(image_size=256, patch_size=16, image_encoder_embedding_planes=128, image_encoder_block_nums=2,
                           image_encoder_head_nums=2, image_encoder_window_size=7, image_encoder_global_attn_indexes=[1])
    config.train_criterion = losses.SAMLoss()
    config.train_dataset = SyntheticSamSet()
    config.batch_size = 2
    config.accumulation_steps = 1
    config.mask_out_idxs = [0, 1, 2, 3]
    config.mask_threshold = 0.0
    config.decoder_iters = 2
    config.use_single_prompt = True
    config.prompt_probs = {'prompt_point': 0.5, 'prompt_box': 0.5, 'prompt_mask': 0.}
    config.frozen_image_encoder = config.frozen_prompt_encoder = config.frozen_mask_decoder = False
    config.optimizer = ('AdamW', {'lr': 1e-4, 'global_weight_decay': False, 'weight_decay': 0,
                                  'no_weight_decay_layer_name_list': []})
    config.scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [100]})
    config.epochs = 1
    config.print_interval = 2
    config.use_amp = True
    config.clip_max_norm = 1.
    config.find_unused_parameters = True
    config.local_rank = 0
    config.group = None
    config.gpus_num = 1
    config.sync_bn = False
    model = config.model.cuda()
    optimizer, _ = utils.build_optimizer(config, model)
    scheduler = utils.Scheduler(config, optimizer)
    model, _, config.scaler = utils.build_training_mode(config, model)
    before = model.arena.flat_param.clone()
    loader = torch.utils.data.DataLoader(config.train_dataset, batch_size=2, shuffle=False, drop_last=True,
                                         collate_fn=SAMBatchCollater(resize=256))
    batch = next(iter(loader))
    assert batch['image'].shape == (2, 3, 256, 256) and batch['image'].is_contiguous()
    assert batch['mask'].shape == (2, 1, 256, 256) and batch['prompt_mask'].shape == (2, 1, 64, 64)
    assert float(batch['image'][:, :, :, 224:].abs().sum()) == 0.0          # zero-padded canvas
    logger = logging.getLogger('saicv_test_sam')
    logger.setLevel(logging.INFO)
    with caplog.at_level(logging.INFO, logger='saicv_test_sam'):
        l1 = iss.train_sam_segmentation(loader, model, config.train_criterion, optimizer, scheduler, 1, logger, config)
    assert l1 > 0 and l1 == l1
    assert 'train: epoch 0001, iter [000002, 000004]' in caplog.text and 'focal_loss:' in caplog.text
    assert 'skip this batch!' not in caplog.text
    assert not torch.equal(before, model.arena.flat_param)
    assert torch.isfinite(model.arena.flat_param).all()


def test_compute_macs_and_params_counts_the_matrix_products():
    """tools.utils.compute_macs_and_params (reference tools/utils.py:119-142, calflops there): the engine's own accounting on one
    eval forward.  Known values: ResNet-50 at 224 has 25 557 032 parameters and 4.089 G multiply-accumulates in its convolutions
    + classifier (torchvision's published 4.09 GFLOPs-as-MACs figure); ViT-B/16 at 224 has 86 567 656 parameters and about
    17.5 GMACs (projections + MLPs 16.8, attention products 0.36 x 2, patch embedding 0.116)."""
    from types import SimpleNamespace
    from simpleaicv_pytorch_training_examples_amd import ops
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones
    from simpleaicv_pytorch_training_examples_amd.tools.utils import compute_macs_and_params
    torch.manual_seed(0)
    model = backbones.resnet50(num_classes=1000).train()
    flops, macs, params = compute_macs_and_params(SimpleNamespace(input_image_size=224), model)
    assert params == '25.557 M' and macs.endswith(' GMACs') and flops.endswith(' GFLOPS'), (flops, macs, params)
    assert abs(float(macs.split()[0]) - 4.089) < 0.02 and abs(float(flops.split()[0]) - 2 * 4.089) < 0.04, (flops, macs)
    assert model.training and next(model.parameters()).is_cuda and not ops.KernelTimer.enabled and ops.KernelTimer.records == []
    vit = backbones.vit_base_patch16(image_size=224, num_classes=1000)
    flops, macs, params = compute_macs_and_params(SimpleNamespace(input_image_size=[224, 224]), vit)
    assert params == '86.568 M', params
    assert 17.2 < float(macs.split()[0]) < 17.8 and macs.endswith(' GMACs'), macs
    print(f'compute_macs_and_params: resnet50 ok, vit_base_patch16 {flops} / {macs} / {params}')
