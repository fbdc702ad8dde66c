"""BASELINE.json configs[0] on the reference's CPU path: ONE EPOCH (781 iterations of batch 64) of the REFERENCE loop
(tools/scripts.py:116-275 train_classification) driving the reference resnet18cifar / CELoss / build_optimizer (torch.optim.SGD)
/ Scheduler, fed by the reference CIFAR100Dataset (SimpleAICV/classification/datasets/cifar100dataset.py) reading CIFAR-100-format
pickles and the reference ClassificationCollater -- on the host cores, fp32, no GPU.  TEST / MEASUREMENT INFRASTRUCTURE.

    python oracle/run_reference_cifar_epoch.py [log path]        (build container only: imports /root/reference)

There is no network, so the pickles are synthetic (50 000 x 3072 uint8 + 100 fine labels, written to a temporary directory in
the CIFAR-100 python format); torchvision / PIL are not in the image, so the config's Opencv2PIL -> pad / flip / crop -> ToTensor
chain is replaced by the one transform that matters to the loop's arithmetic (mean / std normalisation, same constants as
cifar100/resnet18cifar/train_config.py:55-60).  As in oracle/make_golden_traj.py the loop's `.cuda()` calls are made identity,
the per-iteration barrier a no-op (single gloo rank) and get_amp_type a constant; nothing else of the reference is touched.
The log (the loop's own logger lines + wall time per 50 iterations) goes to profiles/r05_cfg1_reference_cpu_epoch.log; the
engine's run of the SAME experiment (same pickle bytes from scripts/cifar_synthetic_pickles.py, same seed-0 weights, same
DistributedSampler order, fp32) through its entry script is profiles/r05_cfg1_engine_gpu_epoch.log (scripts/gpu_cfg1_r05.sh), and
profiles/r05_cfg1_loss_columns.md puts the two loss columns side by side."""
import logging
import os
import pickle
import sys
import tempfile
import time
import types

import numpy as np
import torch
import torch.distributed as dist

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH, CLASSES = 64, 100


sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from cifar_synthetic_pickles import Normalize, write_pickles  # noqa: E402  (the same bytes the engine's entry script reads)


def main():
    log_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r05_cfg1_reference_cpu_epoch.log')
    sys.path.insert(0, REF)
    for name in ['calflops', 'cv2', 'torchvision', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask',
                 'pycocotools.cocoeval', 'PIL', 'PIL.Image']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    for n in ('Image', 'ImageOps', 'ImageEnhance', 'ImageFilter', 'ImageDraw'):
        setattr(sys.modules['PIL'], n, types.ModuleType('PIL.' + n))
    sys.modules['PIL'].__version__ = '9.0'          # auto_rand_augment.py reads the version and two resampling constants at import
    sys.modules['PIL'].Image.Resampling = types.SimpleNamespace(BILINEAR=2, BICUBIC=3, NEAREST=0)
    sys.modules['PIL'].Image.AFFINE = 0
    sys.modules['calflops'].calculate_flops = lambda *a, **k: None
    sys.modules['pycocotools.cocoeval'].COCOeval = object
    sys.modules['pycocotools'].mask = sys.modules['pycocotools.mask']
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29613', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('gloo', rank=0, world_size=1)
    torch.Tensor.cuda = lambda self, *a, **k: self
    dist.barrier = lambda *a, **k: None
    threads = int(os.environ.get('SAICV_CPU_THREADS', '16'))
    torch.set_num_threads(threads)
    from tools import scripts as S
    from tools import utils as U
    from SimpleAICV.classification import backbones, losses
    from SimpleAICV.classification.common import ClassificationCollater
    from SimpleAICV.classification.datasets.cifar100dataset import CIFAR100Dataset
    S.get_amp_type = lambda model: torch.float16

    class Cfg:
        optimizer = ('SGD', {'lr': 0.1, 'momentum': 0.9, 'global_weight_decay': False, 'weight_decay': 5e-4,
                             'no_weight_decay_layer_name_list': []})
        scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.2, 'milestones': [60, 120, 160]})
        epochs, batch_size, accumulation_steps, print_interval = 200, BATCH, 1, 10
        use_amp = use_ema_model = False
        local_rank, gpus_num, group = 0, 1, None

    logger = logging.getLogger('cfg1')
    logger.setLevel(logging.INFO)
    fh = logging.FileHandler(log_path, mode='w')
    fh.setFormatter(logging.Formatter('%(asctime)s - %(message)s', '%Y-%m-%d %H:%M:%S'))
    logger.handlers = [fh]
    with tempfile.TemporaryDirectory() as d:
        write_pickles(d)
        ds = CIFAR100Dataset(root_dir=d, set_name='train', transform=Normalize())
        # the entry script's own loader (tools/train_classification_model.py:72-82): DistributedSampler(shuffle=True), set_epoch(1)
        # -- the engine's entry script builds the same one, so both runs see the same batches in the same order
        sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True)
        sampler.set_epoch(1)
        loader = torch.utils.data.DataLoader(ds, batch_size=BATCH, shuffle=False, num_workers=4, drop_last=True,
                                             collate_fn=ClassificationCollater(), sampler=sampler)
        torch.manual_seed(0)
        model = backbones.resnet18cifar(num_classes=CLASSES)
        model.no_sync = None
        cfg = Cfg()
        optimizer, _ = U.build_optimizer(cfg, model)
        scheduler = U.Scheduler(cfg, optimizer)
        logger.info(f'reference loop tools/scripts.py train_classification on the host: resnet18cifar, batch {BATCH}, '
                    f'{len(ds) // BATCH} iterations, fp32, {threads} threads of {os.cpu_count()} logical CPUs, synthetic CIFAR-100 pickles')
        t0 = time.time()
        avg = S.train_classification(loader, model, losses.CELoss(), optimizer, scheduler, 1, logger, cfg)
        dt = time.time() - t0
        logger.info(f'epoch 001 done: train_loss {avg:.4f}, {len(ds) // BATCH} iterations in {dt:.1f} s = '
                    f'{(len(ds) // BATCH) * BATCH / dt:.1f} images/s on {threads} host threads')
    dist.destroy_process_group()
    print(open(log_path).read()[-600:])


if __name__ == '__main__':
    main()
