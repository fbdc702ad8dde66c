"""Loop-library surface of the reference's tools/utils.py on the MI355X engine.

Same names, arguments and return values as the reference (file:line cited per function) so a
reference tools/train_*.py only changes its import root:
  get_logger (:66-92), set_seed (:95-107), worker_seed_init_fn (:110-116), EmaModel (:145-172),
  compute_macs_and_params (:119-142), build_training_mode (:175-202), Scheduler (:205-289), build_optimizer (:292-679).
What changes underneath: build_training_mode wraps the model in the flat-arena RCCL engine
(engine.DistributedDataParallel) and returns the sync-free GradScaler; build_optimizer returns
the fused flat SGD / AdamW.  Everything is device agnostic (CPU + gloo works for plumbing runs).
"""
import copy
import logging
import logging.handlers
import math
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from .. import engine


def get_logger(name, log_dir):
    '''rank-agnostic logger: <log_dir>/<name>.info.log (weekly rotation) + stream'''
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    logger.propagate = False
    fmt = logging.Formatter('%(asctime)s - %(message)s', datefmt='%Y-%m-%d %H:%M:%S')
    fh = logging.handlers.TimedRotatingFileHandler(os.path.join(log_dir, f'{name}.info.log'), when='W0',
                                                   encoding='utf-8')
    sh = logging.StreamHandler()
    for h in (fh, sh):
        h.setLevel(logging.INFO)
        h.setFormatter(fmt)
        logger.addHandler(h)
    return logger


def set_seed(seed):
    os.environ['PYTHONHASHSEED'] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    # The reference asks for deterministic kernels here (cudnn.deterministic = True, tools/utils.py:106-107).  This engine's
    # counterpart is ops.set_deterministic(): every reduction of the HIP library is ordered (weight / bias gradients, statistics,
    # table gradients, loss sums, the gradient norm: csrc/det.h) and the convolution epilogues write their BatchNorm statistics as
    # fixed-order partial rows -- an fp32 step is bit-reproducible run to run.  SAICV_DETERMINISTIC=0 keeps the fast atomically
    # accumulated path (bench.py sets it unless --deterministic); SAICV_BN_INLINE=1 keeps the atomic BatchNorm statistics only.
    # There is no autotuning to pin.
    from .. import ops
    if os.environ.get('SAICV_DETERMINISTIC', '1') != '0':
        ops.set_deterministic(True)
        if os.environ.get('SAICV_BN_INLINE') == '1':
            ops.BN_INLINE = True
    elif os.environ.get('SAICV_BN_INLINE') is None:
        ops.BN_INLINE = False


def worker_seed_init_fn(worker_id, num_workers, local_rank, seed):
    # same worker gets a fresh seed each epoch (reference adds the wall clock on purpose)
    worker_seed = num_workers * local_rank + worker_id + seed + int(time.time())
    np.random.seed(worker_seed)
    random.seed(worker_seed)


def _with_unit(value, suffix):
    """calflops' number_to_string form: 4089184256 -> '4.089 G' + suffix (three decimals, trailing zeros dropped)"""
    for mag, unit in ((1e12, 'T'), (1e9, 'G'), (1e6, 'M'), (1e3, 'K')):
        if value >= mag:
            return f'{round(value / mag, 3):g} {unit}{suffix}'
    return f'{round(value, 3):g} {suffix}'


def compute_macs_and_params(config, model):
    """-> (flops, macs, params) as strings, for the test entry scripts' `model: ..., flops: ..., macs: ..., params: ...` line
    (reference tools/utils.py:119-142: calflops on a CPU copy of the model with one random image of config.input_image_size).

    calflops is not in the image and the HIP modules do not run on the CPU; the count comes from the engine's own accounting
    instead: one eval-mode forward of a single random image on the GPU with ops.KernelTimer bracketing every matrix-product launch
    (convolutions incl. depthwise, linears, patch embedding, attention products) -- each site reports its algorithmic flops from
    the LOGICAL shapes (unpadded channels).  macs = flops / 2; normalisation / activation / pooling arithmetic is not counted (on
    ResNet-50 calflops reports 8.21 GFLOPS / 4.09 GMACs, this 8.18 / 4.09).  The model is left on the GPU (the entry scripts move
    it there next anyway) in the mode it came in."""
    from .. import ops
    size = config.input_image_size
    assert isinstance(size, (int, list)), 'Illegal input_image_size type!'
    h, w = (size, size) if isinstance(size, int) else (size[0], size[1])
    params = sum(p.numel() for p in model.parameters())
    was_training = model.training
    model = model.cuda().eval()
    timer = ops.KernelTimer
    saved = (timer.enabled, timer.only, timer.records)
    timer.enabled, timer.only, timer.records = True, None, []
    try:
        with torch.no_grad():
            model(torch.randn(1, 3, h, w, device='cuda'))
        torch.cuda.synchronize()
        flops = float(sum(r[3] for r in timer.records))
    finally:
        timer.enabled, timer.only, timer.records = saved
        model.train(was_training)
    return _with_unit(flops, 'FLOPS'), _with_unit(flops / 2, 'MACs'), _with_unit(params, '')


class EmaModel(nn.Module):
    """ema = decay * ema + (1 - decay) * model over every state_dict entry
    (reference tools/utils.py:145-172)."""

    def __init__(self, model, decay=0.9999):
        super(EmaModel, self).__init__()
        src = model.module if hasattr(model, 'module') else model
        arena = src.__dict__.pop('_saicv_arena', None)       # never deep-copy / alias the arenas
        try:
            self.ema_model = copy.deepcopy(src)
        finally:
            if arena is not None:
                src._saicv_arena = arena
        with torch.no_grad():                                # own storage, same layouts
            for p in self.ema_model.parameters():
                p.data = p.data.clone(memory_format=torch.preserve_format)
                p.grad = None
                p.__dict__.pop('_saicv_direct', None)
        self.ema_model.eval()
        self.decay = decay

    def update(self, model, skip_flag=None):
        """skip_flag (device scalar, 1 = this iteration was skipped on the device): the average does not move, as the
        reference's `continue` leaves it (tools/scripts.py:196-200), without a host read of the flag."""
        src = model.module if hasattr(model, 'module') else model
        dst = self.ema_model.module if hasattr(self.ema_model, 'module') else self.ema_model
        with torch.no_grad():
            ev, mv = [], []
            for e, m in zip(dst.state_dict().values(), src.state_dict().values()):
                assert e.shape == m.shape, 'wrong ema model!'
                if e.dtype.is_floating_point:
                    ev.append(e)
                    mv.append(m.detach())
                else:
                    e.copy_(m)
            if skip_flag is None:
                torch._foreach_lerp_(ev, mv, 1.0 - self.decay)
            else:
                w = ((1.0 - self.decay) * (1.0 - skip_flag.reshape(()).float().clamp(0, 1)))
                torch._foreach_lerp_(ev, mv, [w.to(e.dtype) for e in ev])
        from .. import ops
        ops.bump_weights_epoch()


class _ModuleHolder(nn.Module):
    """`.module` pass-through with DDP-style state_dict keys, for the non-trained EMA copy."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def build_training_mode(config, model):
    """(ddp_model, ema_model, scaler) exactly like the reference, on the MI355X engine."""
    ema_model, scaler = None, None
    if getattr(config, 'sync_bn', False):
        raise NotImplementedError('sync_bn=True is not enabled by any reference config and is not implemented')
    find_unused = getattr(config, 'find_unused_parameters', False)
    local_rank = config.local_rank
    group = getattr(config, 'group', None)
    if getattr(config, 'use_ema_model', False):
        ema_model = EmaModel(model, decay=config.ema_model_decay)
        # the reference wraps the EMA copy in DDP too, so checkpoints carry `module.`-prefixed keys
        # and `config.ema_model.ema_model.module` is the bare network (train_classification_model.py:217)
        ema_model.ema_model = _ModuleHolder(ema_model.ema_model)
    model = engine.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                           find_unused_parameters=find_unused, process_group=group)
    if getattr(config, 'use_amp', False):
        device = next(model.parameters()).device
        scaler = engine.GradScaler(device=device)
    return model, ema_model, scaler


class Scheduler:
    """Per-iteration learning rate on a fractional epoch: linear warm-up, then MultiStepLR /
    CosineLR / PolyLR; every param group keeps its own base lr (reference tools/utils.py:205-289)."""

    def __init__(self, config, optimizer):
        self.scheduler_name = config.scheduler[0]
        self.scheduler_parameters = config.scheduler[1]
        self.warm_up_epochs = self.scheduler_parameters['warm_up_epochs']
        self.epochs = config.epochs
        self.optimizer_parameters = config.optimizer[1]
        self.lr = self.optimizer_parameters['lr']
        self.current_lr = self.lr
        self.init_param_groups_lr = [g['lr'] for g in optimizer.param_groups]
        assert self.scheduler_name in ['MultiStepLR', 'CosineLR', 'PolyLR'], 'Unsupported scheduler!'
        assert self.warm_up_epochs >= 0, 'Illegal warm_up_epochs!'
        assert self.epochs > 0, 'Illegal epochs!'

    def _lr_at(self, epoch, base):
        sp = self.scheduler_parameters
        if epoch < self.warm_up_epochs:
            return epoch / self.warm_up_epochs * base
        if self.scheduler_name == 'MultiStepLR':
            return sp['gamma'] ** len([m for m in sp['milestones'] if m <= epoch]) * base
        min_lr = sp.get('min_lr', 0.)
        progress = (epoch - self.warm_up_epochs) / (self.epochs - self.warm_up_epochs)
        if self.scheduler_name == 'CosineLR':
            return 0.5 * (math.cos(progress * math.pi) + 1) * (base - min_lr) + min_lr
        return ((1 - progress) ** sp['power']) * (base - min_lr) + min_lr

    def step(self, optimizer, epoch):
        assert len(self.init_param_groups_lr) == len(optimizer.param_groups)
        for base, group in zip(self.init_param_groups_lr, optimizer.param_groups):
            group['lr'] = self._lr_at(epoch, base)
        self.current_lr = self._lr_at(epoch, self.lr)

    def state_dict(self):
        return {key: value for key, value in self.__dict__.items()}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)


def _per_parameter_settings(config, model):
    """(name, param, weight_decay, lr, lr_scale) for every trainable parameter, following the
    rules of reference build_optimizer (:292-478): `global_weight_decay=False` zeroes decay on 1-d
    parameters and on names in `no_weight_decay_layer_name_list`; `sub_layer_weight_decay` /
    `sub_layer_lr` override by name substring (first match); ViT layer-wise lr decay applies when
    `block_name` is configured: parameters whose name contains it are split evenly, in order,
    over `lr_layer_decay_block` and scaled by decay**(L - layer), the rest get scale 1 except the
    names in the reference's scale-0 list."""
    op = config.optimizer[1]
    lr, weight_decay = op['lr'], op['weight_decay']
    global_wd = op.get('global_weight_decay', True)
    no_decay = op['no_weight_decay_layer_name_list'] if isinstance(op.get('no_weight_decay_layer_name_list'), list) else []
    sub_wd = op['sub_layer_weight_decay'] if isinstance(op.get('sub_layer_weight_decay'), dict) else {}
    sub_lr = op['sub_layer_lr'] if isinstance(op.get('sub_layer_lr'), dict) else {}
    layer_decay = 'block_name' in op

    def decay_of(name, p):
        if global_wd:
            return weight_decay
        if p.ndim == 1 or any(k in name for k in no_decay):
            return 0.
        return next((v for k, v in sub_wd.items() if k in name), weight_decay)

    def lr_of(name):
        return next((v for k, v in sub_lr.items() if k in name), lr)

    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    if not layer_decay:
        return [(n, p, decay_of(n, p), lr_of(n), None) for n, p in named]

    blocks = op['lr_layer_decay_block']
    num_layers = len(blocks) + 1
    scales = [op['lr_layer_decay'] ** (num_layers - i) for i in range(num_layers + 1)]
    scale0_names = ['position_encoding', 'cls_token', 'patch_embedding']
    outside = [(n, p) for n, p in named if op['block_name'] not in n]
    inside = [(n, p) for n, p in named if op['block_name'] in n]
    out = [(n, p, decay_of(n, p), lr_of(n), scales[0] if any(k in n for k in scale0_names) else 1.) for n, p in outside]
    per_block = len(inside) // len(blocks)
    for layer_id in range(len(blocks)):
        for n, p in inside[layer_id * per_block:(layer_id + 1) * per_block]:
            out.append((n, p, decay_of(n, p), lr_of(n), scales[layer_id + 1]))
    return out


def build_optimizer(config, model):
    """-> (optimizer, model_layer_weight_decay_list), reference tools/utils.py:292-679 (SGD and
    AdamW branches; Muon is outside the hot path).  Parameter groups are the distinct
    (weight_decay, lr[, lr_scale]) combinations in first-appearance order."""
    optimizer_name, op = config.optimizer[0], config.optimizer[1]
    assert optimizer_name in ['SGD', 'AdamW', 'Muon'], 'Unsupported optimizer!'
    if optimizer_name == 'Muon':
        raise NotImplementedError('Muon is only used by the universal-segmentation configs (out of scope)')
    settings = _per_parameter_settings(config, model)
    groups, summary, index = [], [], {}
    for name, p, wd, lr, scale in settings:
        key = (wd, lr, scale)
        if key not in index:
            index[key] = len(groups)
            groups.append({'params': [], 'weight_decay': wd, 'lr': lr * (scale if scale is not None else 1.)})
            entry = {'name': [], 'weight_decay': wd, 'lr': lr}
            if scale is not None:
                entry['lr_scale'] = scale
            summary.append(entry)
        groups[index[key]]['params'].append(p)
        summary[index[key]]['name'].append(name)
    target = model.module if hasattr(model, 'module') else model
    if optimizer_name == 'SGD':
        opt = engine.SGD(target, groups, lr=op['lr'], momentum=op['momentum'], nesterov=op.get('nesterov', False))
    else:
        opt = engine.AdamW(target, groups, lr=op['lr'], betas=(op.get('beta1', 0.9), op.get('beta2', 0.999)),
                           eps=op.get('eps', 1e-08))
    return opt, summary
