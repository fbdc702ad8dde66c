"""MAE pre-training fixtures produced by RUNNING THE REFERENCE (imported from /root/reference).  TEST INFRASTRUCTURE.

Build container only:   python oracle/make_golden_mae.py        -> tests/golden/mae_collater.pt, tests/golden/traj_mae_tiny.pt

mae_collater    reference MAESelfSupervisedPretrainCollater (SimpleAICV/masked_image_modeling/common.py:16-56) on three seeded
                HWC float batches (image 64 / patch 16 with and without norm_label, image 32 / patch 8): 'image' and 'label'.
traj_mae_tiny   the REFERENCE LOOP ITSELF: tools/scripts.py:1774-1934 (train_mae_self_supervised_learning) driving the reference
                VITMAEPretrainModel (the mae_tiny geometry of make_golden_f2.py) / MSELoss / build_optimizer (torch.optim.AdamW,
                lr 1.5e-3, betas (0.9, 0.95), weight decay 0.05 with 1-d parameters at 0) / Scheduler (CosineLR, 2 warm-up epochs
                of 400) on CPU in fp32 for 12 iterations of batch 16; batches come through the reference collater.  The forward's
                only random draw is torch.rand(B, L) in random_masking: torch.manual_seed(77) right before the loop, so iteration i
                uses the i-th draw (the test replays them).  As in make_golden_traj.py the loop's `.cuda()` calls are made identity,
                the per-iteration barrier a no-op and get_amp_type a constant; nothing else of the reference is touched.  The run is
                repeated with another thread count (another fp32 summation order) to record how far the reference moves from itself."""
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
MAE_TINY = dict(patch_size=16, image_size=64, mask_ratio=0.75, encoder_embedding_planes=128, encoder_block_nums=2,
                encoder_head_nums=2, decoder_embedding_planes=64, decoder_block_nums=2, decoder_head_nums=2)
STEPS, BATCH, NOISE_SEED = 12, 16, 77


def samples(seed, n, size):
    rng = np.random.default_rng(seed)
    return [{'image': rng.standard_normal((size, size, 3), dtype=np.float32) * 0.7 + 0.1, 'label': 0} for _ in range(n)]


def collater_case():
    from SimpleAICV.masked_image_modeling.common import MAESelfSupervisedPretrainCollater
    cases = {}
    for key, (size, patch, norm, seed) in {'i64_p16_norm': (64, 16, True, 3), 'i64_p16_raw': (64, 16, False, 4),
                                            'i32_p8_norm': (32, 8, True, 5)}.items():
        out = MAESelfSupervisedPretrainCollater(image_size=size, patch_size=patch, norm_label=norm)(samples(seed, 3, size))
        cases[key] = {'size': size, 'patch': patch, 'norm': norm, 'seed': seed, 'image': out['image'].clone(),
                      'image_stride': tuple(out['image'].stride()), 'label': out['label'].clone()}
        print(key, tuple(out['image'].shape), tuple(out['image'].stride()), tuple(out['label'].shape), float(out['label'].abs().mean()))
    torch.save({'cases': cases}, os.path.join(OUT, 'mae_collater.pt'))


class Cfg:
    pass


def make_config():
    c = Cfg()
    c.optimizer = ('AdamW', {'lr': 1.5e-3, 'global_weight_decay': False, 'weight_decay': 5e-2,
                             'no_weight_decay_layer_name_list': [], 'beta1': 0.9, 'beta2': 0.95})
    c.scheduler = ('CosineLR', {'warm_up_epochs': 2, 'min_lr': 1e-6})
    c.epochs = 400
    c.batch_size = BATCH
    c.accumulation_steps = 1
    c.print_interval = 4
    c.use_amp = False
    c.use_ema_model = False
    c.local_rank = 0
    c.gpus_num = 1
    c.group = None
    return c


class Loader(list):
    def __init__(self, items):
        super().__init__(items)
        self.dataset = [None] * (len(items) * BATCH)


def run(threads):
    from tools import scripts as S
    from tools import utils as U
    from SimpleAICV.masked_image_modeling.models.vit_mae import VITMAEPretrainModel
    from SimpleAICV.masked_image_modeling.losses import MSELoss
    from SimpleAICV.masked_image_modeling.common import MAESelfSupervisedPretrainCollater
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = VITMAEPretrainModel(**MAE_TINY)
    model.no_sync = None
    cfg = make_config()
    optimizer, _ = U.build_optimizer(cfg, model)
    scheduler = U.Scheduler(cfg, optimizer)
    coll = MAESelfSupervisedPretrainCollater(image_size=64, patch_size=16, norm_label=True)
    batches = [coll(samples(1000 + i, BATCH, 64)) for i in range(STEPS)]
    trace, step_losses = [], []

    class Rec(logging.Handler):
        def emit(self, record):
            trace.append(record.getMessage())

    logger = logging.getLogger('traj_mae')
    logger.setLevel(logging.INFO)
    logger.handlers = [Rec()]
    orig_update = S.AverageMeter.update

    def spy(self, val, n=1):
        step_losses.append(float(val))
        return orig_update(self, val, n)

    S.AverageMeter.update = spy
    S.get_amp_type = lambda model: torch.float16      # queries the GPU's compute capability; unused with use_amp=False
    torch.manual_seed(NOISE_SEED)
    try:
        avg = S.train_mae_self_supervised_learning(Loader(batches), model, MSELoss(), optimizer, scheduler, 1, logger, cfg)
    finally:
        S.AverageMeter.update = orig_update
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return {'losses': step_losses, 'avg_loss': float(avg), 'log': trace, 'lr': scheduler.current_lr,
            'param_norms': {k: float(v.float().norm()) for k, v in sd.items() if v.dtype.is_floating_point},
            'final_state': {k: sd[k] for k in ('encoder.patch_embed.proj.weight', 'encoder.cls_token', 'decoder.mask_token',
                                               'encoder_to_decoder.weight', 'decoder.fc.weight') if k in sd}}


def traj_case():
    a = run(8)
    b = run(3)
    noise = [abs(x - y) / abs(x) for x, y in zip(a['losses'], b['losses'])]
    a['reference_noise'] = {'loss_rel': noise,
                            'param_norm_rel': max(abs(a['param_norms'][k] - b['param_norms'][k]) / max(a['param_norms'][k], 1e-12)
                                                  for k in a['param_norms'])}
    c = make_config()
    a['config'] = {'steps': STEPS, 'batch': BATCH, 'kwargs': MAE_TINY, 'model_seed': 0, 'noise_seed': NOISE_SEED, 'data_seed0': 1000,
                   'optimizer': c.optimizer, 'scheduler': c.scheduler, 'epochs': c.epochs}
    print('losses', [round(v, 5) for v in a['losses']])
    print('noise ', [f'{v:.1e}' for v in noise], 'param norm noise', a['reference_noise']['param_norm_rel'])
    print('log:', a['log'][:3], 'state keys', list(a['final_state']))
    torch.save(a, os.path.join(OUT, 'traj_mae_tiny.pt'))
    print('wrote traj_mae_tiny.pt', os.path.getsize(os.path.join(OUT, 'traj_mae_tiny.pt')) // 1024, 'KiB')


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    sys.path.insert(0, REF)
    for name in ['calflops', 'cv2', 'torchvision', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['calflops'].calculate_flops = lambda *a, **k: None
    sys.modules['pycocotools.cocoeval'].COCOeval = object
    sys.modules['pycocotools'].mask = sys.modules['pycocotools.mask']
    only = set(sys.argv[1:])
    if not only or 'mae_collater' in only:
        collater_case()
    if not only or 'traj_mae_tiny' in only:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29613', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('gloo', rank=0, world_size=1)
        torch.Tensor.cuda = lambda self, *a, **k: self             # the loop's images.cuda() / labels.cuda()
        dist.barrier = lambda *a, **k: None                        # barrier(device_ids=[local_rank]) needs a GPU
        traj_case()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
