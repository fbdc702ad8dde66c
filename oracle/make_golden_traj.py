"""Loss-trajectory fixture produced by the REFERENCE LOOP ITSELF: tools/scripts.py:116-275 (train_classification)
driving the reference resnet18cifar / CELoss / tools.utils.build_optimizer (torch.optim.SGD) / Scheduler on CPU in
fp32 for 20 iterations of batch 64 (BASELINE.json configs[0] shapes: 3x32x32, 100 classes, SGD lr 0.1 momentum 0.9
weight decay 5e-4 with 1-d parameters at 0, cifar100/resnet18cifar/train_config.py:67-87).  Run in the build container:
    python oracle/make_golden_traj.py   ->  tests/golden/traj_resnet18cifar_b64.pt
The loop's hard `.cuda()` calls are made identity, the per-iteration barrier a no-op (single gloo rank) and the GPU
capability query of get_amp_type (unused without AMP) a constant; nothing
else of the reference is touched.  The run is repeated NINE more times under other fp32 summation orders for the same mathematics
(true-NCHW or NHWC-strided inputs x 1 / 2 / 3 / 5 threads, and both layouts with the oneDNN convolutions switched off: other ATen
kernels, other reduction splits) to record how far the reference moves from ITSELF per iteration: training amplifies rounding
differences, and the parity gate of tests/test_gpu_trajectories.py is set from the per-iteration MAXIMUM over those runs
(`reference_envelope`; r05's single perturbed run, `reference_noise`, was one sample of it and is kept for comparison)."""
import logging
import os
import sys
import types

import torch
import torch.distributed as dist

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'traj_resnet18cifar_b64.pt')
STEPS, BATCH, CLASSES = 20, 64, 100


class Cfg:
    pass


def make_config(lr=0.1):
    c = Cfg()
    c.optimizer = ('SGD', {'lr': lr, 'momentum': 0.9, 'global_weight_decay': False, 'weight_decay': 5e-4,
                           'no_weight_decay_layer_name_list': []})
    c.scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.2, 'milestones': [60, 120, 160]})
    c.epochs = 200
    c.batch_size = BATCH
    c.accumulation_steps = 1
    c.print_interval = 5
    c.use_amp = False
    c.use_ema_model = False
    c.local_rank = 0
    c.gpus_num = 1
    c.group = None
    return c


def batches(nchw=False):
    g = torch.Generator().manual_seed(123)
    out = []
    for _ in range(STEPS):
        x = torch.randn(BATCH, 32, 32, 3, generator=g).permute(0, 3, 1, 2)     # the collater's NHWC-strided NCHW view
        y = torch.randint(0, CLASSES, (BATCH,), generator=g)
        out.append({'image': x.contiguous() if nchw else x, 'label': y})
    return out


class Loader(list):
    """len(loader.dataset) // batch_size = iterations per epoch (reference scripts.py:137)"""

    def __init__(self, items):
        super().__init__(items)
        self.dataset = [None] * (len(items) * BATCH)


# (input layout, threads, oneDNN convolutions) of the perturbed runs; the base run is (NHWC-strided, 8, on)
VARIANTS = [(True, 3, True), (True, 1, True), (True, 2, True), (True, 5, True), (False, 1, True), (False, 3, True), (False, 5, True),
            (True, 4, False), (False, 8, False)]


def run(nchw, lr=0.1, threads=None, mkldnn=True, double=False):
    from tools import scripts as S
    from tools import utils as U
    from SimpleAICV.classification import backbones, losses
    torch.manual_seed(0)
    model = backbones.resnet18cifar(num_classes=CLASSES)
    if double:          # the arbiter: the same loop in float64 (weights drawn in fp32 as everywhere, then widened)
        model = model.double()
    torch.set_num_threads(threads if threads is not None else (3 if nchw else 8))
    torch.backends.mkldnn.enabled = mkldnn
    model.no_sync = None
    cfg = make_config(lr)
    optimizer, _ = U.build_optimizer(cfg, model)
    scheduler = U.Scheduler(cfg, optimizer)
    crit = losses.CELoss()
    trace = []

    class Rec(logging.Handler):
        def emit(self, record):
            trace.append(record.getMessage())

    logger = logging.getLogger('traj')
    logger.setLevel(logging.INFO)
    logger.handlers = [Rec()]
    step_losses = []
    orig_update = S.AverageMeter.update

    def spy(self, val, n=1):
        step_losses.append(float(val))
        return orig_update(self, val, n)

    S.AverageMeter.update = spy
    S.get_amp_type = lambda model: torch.float16      # queries the GPU's compute capability; unused with use_amp=False
    try:
        data = batches(nchw)
        if double:
            data = [{'image': d['image'].double(), 'label': d['label']} for d in data]
        avg = S.train_classification(Loader(data), model, crit, optimizer, scheduler, 1, logger, cfg)
    finally:
        S.AverageMeter.update = orig_update
    model.eval()
    with torch.no_grad():
        probe = model(batches()[0]['image'].double() if double else batches()[0]['image'])
    sd = {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in model.state_dict().items()}
    return {'losses': step_losses, 'avg_loss': float(avg), 'log': trace, 'eval_logits': probe,
            'final_state': {k: sd[k] for k in ('conv1.layer.0.weight', 'fc.weight', 'fc.bias', 'conv1.layer.1.running_mean',
                                               'layer4.1.conv2.layer.1.running_var')},
            'param_norms': {k: float(v.float().norm()) for k, v in sd.items() if v.dtype.is_floating_point},
            'lr': scheduler.current_lr}


def main():
    sys.path.insert(0, REF)
    for name in ['calflops', 'cv2', 'torchvision', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask',
                 'pycocotools.cocoeval']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['calflops'].calculate_flops = lambda *a, **k: None
    sys.modules['pycocotools.cocoeval'].COCOeval = object
    sys.modules['pycocotools'].mask = sys.modules['pycocotools.mask']
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('gloo', rank=0, world_size=1)
    torch.Tensor.cuda = lambda self, *a, **k: self             # the loop's images.cuda() / labels.cuda()
    dist.barrier = lambda *a, **k: None                        # barrier(device_ids=[local_rank]) needs a GPU
    torch.set_num_threads(8)
    out = {}
    if '--float64-only' in sys.argv:      # add / refresh the float64 arbiter of an existing fixture (the fp32 runs take 40 minutes)
        out = torch.load(OUT, weights_only=False)
        for lr in (0.1, 0.01):
            a = out[f'lr{lr}']
            d = run(False, lr, 8, True, double=True)
            f64 = d['losses']
            runs = [a['losses']] + a['reference_envelope']['losses']
            a['float64'] = {'losses': f64, 'avg_loss': d['avg_loss'], 'fp32_runs_rel': [[abs(x - y) / abs(y) for x, y in zip(r, f64)] for r in runs]}
            print('lr', lr, 'float64 losses', [round(v, 4) for v in f64])
            for r in a['float64']['fp32_runs_rel']:
                print('  fp32 run vs float64', [f'{v:.1e}' for v in r])
        torch.save(out, OUT)
        print('wrote', OUT, os.path.getsize(OUT) // 1024, 'KiB')
        dist.destroy_process_group()
        return
    for lr in (0.1, 0.01):          # the config's lr (the loss climbs: chaotic) and a tame one (tight comparison)
        a = run(False, lr)
        others = [run(nchw, lr, threads, mk) for nchw, threads, mk in VARIANTS]
        torch.backends.mkldnn.enabled = True
        b = others[0]
        noise = [abs(x - y) / abs(x) for x, y in zip(a['losses'], b['losses'])]
        per_run = [[abs(x - y) / abs(x) for x, y in zip(a['losses'], o['losses'])] for o in others]
        # float64 arbiter: how far each fp32 run of the reference is from the same loop in double precision, per iteration
        d = run(False, lr, 8, True, double=True)
        f64 = d['losses']
        a['float64'] = {'losses': f64, 'avg_loss': d['avg_loss'],
                        'fp32_runs_rel': [[abs(x - y) / abs(y) for x, y in zip(o['losses'], f64)] for o in [a] + others]}
        print('fp32 base vs float64', [f'{v:.1e}' for v in a['float64']['fp32_runs_rel'][0]])
        a['reference_envelope'] = {'loss_rel': [max(r[i] for r in per_run) for i in range(STEPS)], 'per_run': per_run,
                                   'variants': [list(v) for v in VARIANTS], 'losses': [o['losses'] for o in others],
                                   'avg_loss_rel': max(abs(o['avg_loss'] - a['avg_loss']) / a['avg_loss'] for o in others)}
        a['reference_noise'] = {'loss_rel': noise,
                                'eval_logits_rel': float((a['eval_logits'] - b['eval_logits']).abs().max() / a['eval_logits'].abs().max()),
                                'param_norm_rel': max(abs(a['param_norms'][k] - b['param_norms'][k]) / max(a['param_norms'][k], 1e-12)
                                                      for k in a['param_norms'])}
        a['config'] = {'steps': STEPS, 'batch': BATCH, 'classes': CLASSES, 'data_seed': 123, 'model_seed': 0,
                       'optimizer': make_config(lr).optimizer, 'scheduler': make_config(lr).scheduler, 'epochs': 200}
        out[f'lr{lr}'] = a
        print('lr', lr, 'losses', [round(v, 4) for v in a['losses']])
        print('noise ', [f'{v:.1e}' for v in noise])
        print('envelope', [f'{v:.1e}' for v in a['reference_envelope']['loss_rel']])
        print('eval logits noise', a['reference_noise']['eval_logits_rel'], 'param norm noise', a['reference_noise']['param_norm_rel'])
        print('log:', a['log'][:2])
    torch.save(out, OUT)
    print('wrote', OUT, os.path.getsize(OUT) // 1024, 'KiB')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
