"""Loss-trajectory fixtures of the detection and interactive-segmentation LOOPS, produced by the reference loops themselves:
tools/scripts.py:900-1092 (train_detection) and tools/interactive_segmentation_scripts.py:274-564 (train_sam_segmentation),
driving the reference models / losses / tools.utils.build_optimizer (torch.optim.AdamW) / Scheduler on CPU in fp32.
Run in the build container:
    python oracle/make_golden_traj_det_sam.py   ->  tests/golden/traj_detr_r18_tiny.pt, tests/golden/traj_sam_tiny.pt

DETR  : resnet18_detr (20 classes, 20 queries, dropout 0), batch 4 of 192 x 256 images on a 256 x 256 canvas, AdamW 1e-4,
        clip_max_norm 0.1, 8 iterations.
SAM   : the tiny SAM of oracle/make_golden_sam.py, batch 2 of 256 x 256 images, AdamW 1e-4, 6 iterations, two prompt regimes:
        'all'   point + box + mask prompts, one decoder pass (prompt_probs 1 / 1 / 1 -> decoder_iters 0);
        'iters' point + box, then TWO more decoder passes whose click comes from the previous prediction.  The reference draws
                that click with torch.rand (sample_random_point); a random stream cannot be shared with the HIP sampler
                (tests/test_gpu_input.py tests it statistically), so for THIS fixture both sides use the deterministic rule
                `first_error_click` below -- everything else of the iteration logic is the reference's.
As in oracle/make_golden_traj.py: `.cuda()` is the identity, the per-iteration barrier / per-parameter all-reduce of a
one-rank gloo group are no-ops, and each run is repeated with another thread count to record how far the reference moves
from ITSELF per iteration (the gate of the tests)."""
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

DETR_STEPS, DETR_BATCH = 8, 4
SAM_STEPS, SAM_BATCH = 6, 2


def first_error_click(gt_masks, pred_masks, num_pt=1):
    """Deterministic stand-in for sample_random_point (same signature and output contract, reference :202-228): the FIRST
    pixel in row-major order of the error region (label 1 if it is a missed foreground pixel, 0 if falsely predicted), the
    first background pixel (label 0) when the prediction is exact, pixel 0 otherwise."""
    gt = gt_masks.bool()
    pred = torch.zeros_like(gt) if pred_masks is None else pred_masks.bool()
    b, _, h, w = gt.shape
    out = torch.zeros(b, 1, 3)
    for i in range(b):
        g, p = gt[i, 0].flatten(), pred[i, 0].flatten()
        err = g != p
        if bool(err.any()):
            k = int(torch.nonzero(err)[0])
            lab = 1.0 if bool(g[k]) else 0.0
        elif bool((~g).any()):
            k, lab = int(torch.nonzero(~g)[0]), 0.0
        else:
            k, lab = 0, 0.0
        out[i, 0] = torch.tensor([float(k % w), float(k // w), lab])
    return out.to(gt_masks.device)


class Cfg:
    pass


def base_config(batch):
    c = Cfg()
    c.optimizer = ('AdamW', {'lr': 1e-4, 'global_weight_decay': False, 'weight_decay': 1e-4, 'no_weight_decay_layer_name_list': []})
    c.scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [100]})
    c.epochs = 1
    c.batch_size = batch
    c.accumulation_steps = 1
    c.print_interval = 1
    c.use_amp = False
    c.use_ema_model = False
    c.local_rank = 0
    c.gpus_num = 1
    c.group = None
    c.clip_max_norm = 0.1
    return c


class Loader(list):
    def __init__(self, items, batch):
        super().__init__(items)
        self.dataset = [None] * (len(items) * batch)


def _spy_losses(module):
    got = []
    orig = module.AverageMeter.update

    def spy(self, val, n=1):
        got.append(float(val))
        return orig(self, val, n)

    module.AverageMeter.update = spy
    return got, lambda: setattr(module.AverageMeter, 'update', orig)


def detr_batches():
    from oracle.make_golden_detr import detr_inputs
    out = []
    for s in range(DETR_STEPS):
        images, masks, annots = detr_inputs(DETR_BATCH, 1000 + s)
        out.append({'image': images, 'annots': annots, 'scaled_annots': annots, 'mask': masks})
    return out


def run_detr(threads):
    from tools import scripts as S
    from tools import utils as U
    from SimpleAICV.detection.models import detr
    from SimpleAICV.detection import losses
    from oracle.make_golden_detr import zero_dropout
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = detr.resnet18_detr(hidden_inplanes=256, query_nums=20, num_classes=20)
    zero_dropout(model)
    model.no_sync = None
    cfg = base_config(DETR_BATCH)
    cfg.network = 'resnet18_detr'
    optimizer, _ = U.build_optimizer(cfg, model)
    scheduler = U.Scheduler(cfg, optimizer)
    crit = losses.DETRLoss(num_classes=20)
    logger = logging.getLogger('traj_detr')
    logger.handlers = [logging.NullHandler()]
    got, restore = _spy_losses(S)
    S.get_amp_type = lambda model: torch.float16
    try:
        avg = S.train_detection(Loader(detr_batches(), DETR_BATCH), model, crit, optimizer, scheduler, 1, logger, cfg)
    finally:
        restore()
    sd = model.state_dict()
    return {'losses': got, 'avg_loss': float(avg), 'lr': scheduler.current_lr,
            'param_norms': {k: float(v.float().norm()) for k, v in sd.items() if v.dtype.is_floating_point}}


def sam_batches():
    from oracle.make_golden_sam import SAM_TINY, sam_inputs
    out = []
    for s in range(SAM_STEPS):
        images, masks, points, boxes = sam_inputs(SAM_TINY, SAM_BATCH, 2000 + s)
        q = SAM_TINY['image_size'] // 4
        pm = torch.nn.functional.interpolate(masks, size=(q, q), mode='nearest')
        out.append({'image': images, 'mask': masks, 'prompt_point': points, 'prompt_box': boxes, 'prompt_mask': pm})
    return out


class Wrapped(torch.nn.Module):
    """What the reference loop expects of its DDP-wrapped model: `.module`, `no_sync`, parameters through the wrapper."""

    def __init__(self, m):
        super().__init__()
        self.module = m

    def no_sync(self):
        import contextlib
        return contextlib.nullcontext()


def sam_config(regime):
    cfg = base_config(SAM_BATCH)
    cfg.frozen_image_encoder = cfg.frozen_prompt_encoder = cfg.frozen_mask_decoder = False
    cfg.use_single_prompt = False
    cfg.mask_out_idxs = [0, 1, 2, 3]
    cfg.input_image_size = 256
    cfg.mask_threshold = 0.0
    if regime == 'all':
        cfg.prompt_probs = {'prompt_point': 1.0, 'prompt_box': 1.0, 'prompt_mask': 1.0}
        cfg.decoder_iters = 2           # overridden to 0 by the loop because a mask prompt is used
    else:
        cfg.prompt_probs = {'prompt_point': 1.0, 'prompt_box': 1.0, 'prompt_mask': 0.0}
        cfg.decoder_iters = 2
    return cfg


def run_sam(threads, regime):
    from tools import interactive_segmentation_scripts as S
    from tools import utils as U
    from SimpleAICV.interactive_segmentation.models.segment_anything import sam
    from SimpleAICV.interactive_segmentation import losses
    from oracle.make_golden_sam import SAM_TINY
    from oracle.torch_oracle import sam_randomize_zero_init
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    np.random.seed(0)
    net = sam.SAM(**SAM_TINY)
    sam_randomize_zero_init(net.named_parameters(), 100)
    model = Wrapped(net)
    cfg = sam_config(regime)
    optimizer, _ = U.build_optimizer(cfg, model)
    scheduler = U.Scheduler(cfg, optimizer)
    crit = losses.SAMLoss(alpha=0.25, gamma=2, focal_loss_weight=20, dice_loss_weight=1, iou_predict_loss_weight=1,
                          supervise_all_iou=True, mask_threshold=0.0)
    logger = logging.getLogger('traj_sam')
    logger.handlers = [logging.NullHandler()]
    got, restore = _spy_losses(S)
    S.get_amp_type = lambda model: torch.float16
    S.sample_random_point = first_error_click
    try:
        avg = S.train_sam_segmentation(Loader(sam_batches(), SAM_BATCH), model, crit, optimizer, scheduler, 1, logger, cfg)
    finally:
        restore()
    sd = net.state_dict()
    return {'losses': got, 'avg_loss': float(avg), 'lr': scheduler.current_lr,
            'param_norms': {k: float(v.float().norm()) for k, v in sd.items() if v.dtype.is_floating_point}}


def with_noise(a, b):
    a['reference_noise'] = {'loss_rel': [abs(x - y) / abs(x) for x, y in zip(a['losses'], b['losses'])],
                            'param_norm_rel': max(abs(a['param_norms'][k] - b['param_norms'][k]) / max(a['param_norms'][k], 1e-12)
                                                  for k in a['param_norms'])}
    return a


def main():
    sys.path.insert(0, REF)
    for name in ['calflops', 'cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask',
                 'pycocotools.cocoeval', 'pycocotools.coco']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['calflops'].calculate_flops = lambda *a, **k: None
    sys.modules['pycocotools.cocoeval'].COCOeval = object
    sys.modules['pycocotools'].mask = sys.modules['pycocotools.mask']
    # the reference's `tools` is a namespace package (no __init__.py): the repository root, whose `tools` / `SimpleAICV` alias
    # packages would win over it, joins sys.path only AFTER every reference module this script drives is imported
    import tools.scripts, tools.utils, tools.interactive_segmentation_scripts                            # noqa: E401,F401
    import SimpleAICV.detection.models.detr, SimpleAICV.detection.losses                                  # noqa: E401,F401
    import SimpleAICV.interactive_segmentation.models.segment_anything.sam, SimpleAICV.interactive_segmentation.losses  # noqa
    assert tools.utils.__file__.startswith(REF) and SimpleAICV.detection.losses.__file__.startswith(REF)
    sys.path.append(ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29613', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('gloo', rank=0, world_size=1)
    torch.Tensor.cuda = lambda self, *a, **k: self
    dist.barrier = lambda *a, **k: None
    dist.all_reduce = lambda t, *a, **k: None          # one rank: SUM / AVG over the group is the identity (gloo has no AVG)
    only = sys.argv[1:]
    if not only or 'detr' in only:
        fx = with_noise(run_detr(8), run_detr(3))
        fx['config'] = {'steps': DETR_STEPS, 'batch': DETR_BATCH}
        torch.save(fx, os.path.join(OUT, 'traj_detr_r18_tiny.pt'))
        print('detr losses', [round(v, 4) for v in fx['losses']], 'noise', [f'{v:.1e}' for v in fx['reference_noise']['loss_rel']])
    if not only or 'sam' in only:
        out = {}
        for regime in ('all', 'iters'):
            fx = with_noise(run_sam(8, regime), run_sam(3, regime))
            fx['config'] = {'steps': SAM_STEPS, 'batch': SAM_BATCH, 'regime': regime}
            out[regime] = fx
            print('sam', regime, 'losses', [round(v, 4) for v in fx['losses']], 'noise',
                  [f'{v:.1e}' for v in fx['reference_noise']['loss_rel']])
        torch.save(out, os.path.join(OUT, 'traj_sam_tiny.pt'))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
