"""Per-iteration train / eval loops of the reference tools/scripts.py on the MI355X engine.

  all_reduce_operation_in_group_for_variables  (reference :26-33)
  test_classification                          (reference :36-113)
  train_classification                         (reference :116-275)
  compute_voc_ap / compute_ious / evaluate_voc_detection / test_detection   (reference :503-739, :884-897; the COCO
                                               variant needs pycocotools, which the bench image does not have)
  train_detection                              (reference :900-1092)
  train_mae_self_supervised_learning           (reference :1774-1934)

Loop semantics are kept -- skip a batch when ANY rank saw inf/nan input or a zero/inf/nan loss,
gradient accumulation with `no_sync()`, optional clipping, GradScaler step/update, EMA, mean-
over-ranks loss meter, per-iteration scheduler on the fractional epoch, identical log lines --
but the reference's four host synchronisations and the barrier per iteration (C5-C7 in
SURVEY.md section 2.4) collapse into ONE 2-element all-reduce whose result the host reads a few
iterations late:
  * the skip decision is taken on the device: the flag is all-reduced and handed to the fused
    optimizer kernel as `found_inf`, which then leaves parameters and optimizer state untouched
    (what `optimizer.zero_grad(); continue` achieves in the reference);
  * the loss meter, the "skip this batch!" log line and the iteration counter are updated when
    the (flag, loss) pair of an iteration is popped from a short queue (`config.host_sync_lag`
    iterations later, default 2; 0 restores a blocking read every iteration), so the host never
    stalls the HIP stream.
"""
import collections
import time

import numpy as np

import os

import torch
import torch.distributed as dist
from torch.amp.autocast_mode import autocast

from ..engine import any_nonfinite
from ..SimpleAICV.classification.common import AccMeter, AverageMeter, get_amp_type
from ..SimpleAICV.detection.common import pad_mask_on_device


def _device_of(model):
    return next(model.parameters()).device


def _dist_on(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def all_reduce_sum_packed(packed, model, group):
    """SUM over the ranks of a small fp32 device tensor (skip flag + loss terms), in place.  With the engine's DDP wrapper on
    RCCL it travels on the SAME communicator and communication stream as the gradient buckets (saicv_comm_*), so the step
    never has collectives of two communicators in flight at once; otherwise torch.distributed carries it."""
    if not _dist_on(group):
        return packed
    comm = getattr(model, 'comm', None)
    if comm is not None and packed.dtype == torch.float32 and packed.is_contiguous():
        comm.allreduce_now(packed, average=False)
    else:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed


def all_reduce_operation_in_group_for_variables(variables, operator, group):
    """python scalars / 0-d tensors -> all-reduced python scalars (blocking; eval path only)."""
    device = 'cuda' if torch.cuda.is_available() else 'cpu'
    for i in range(len(variables)):
        if not torch.is_tensor(variables[i]):
            variables[i] = torch.tensor(variables[i], device=device)
        if _dist_on(group):
            dist.all_reduce(variables[i], op=operator, group=group)
        variables[i] = variables[i].item()
    return variables


def test_classification(test_loader, model, criterion, config):
    batch_time, data_time, losses, accs = AverageMeter(), AverageMeter(), AverageMeter(), AccMeter()
    if getattr(config, 'use_ema_model', False):
        model = config.ema_model.ema_model
    model.eval()
    device = _device_of(model)
    sync = torch.cuda.synchronize if device.type == 'cuda' else (lambda: None)
    with torch.no_grad():
        end = time.time()
        for data in test_loader:
            images, labels = data['image'].to(device), data['label'].to(device)
            sync()
            data_time.update(time.time() - end)
            end = time.time()
            outputs = model(images)
            sync()
            batch_time.update(time.time() - end)
            loss = criterion(outputs, labels)
            _, topk = torch.topk(outputs, k=5, dim=1, largest=True, sorted=True)
            correct = topk.eq(labels.unsqueeze(-1).expand_as(topk)).float()
            # one fused all-reduce instead of the reference's 1 + 3 scalar ones (C8)
            packed = torch.stack([loss.float(), correct[:, :1].sum(), correct[:, :5].sum(),
                                  torch.tensor(float(images.size(0)), device=device)])
            all_reduce_sum_packed(packed, model, config.group)
            loss_sum, acc1_n, acc5_n, n = packed.tolist()
            losses.update(loss_sum / float(config.gpus_num), images.size(0))
            accs.update(acc1_n, acc5_n, n)
            end = time.time()
    accs.compute()
    per_gpu = config.batch_size // config.gpus_num
    return (accs.acc1 * 100, accs.acc5 * 100, losses.avg, data_time.avg / per_gpu * 1000,
            batch_time.avg / per_gpu * 1000)


def train_classification(train_loader, model, criterion, optimizer, scheduler, epoch, logger, config):
    '''train classification model for one epoch'''
    losses = AverageMeter()
    model.train()
    device = _device_of(model)
    amp_type = get_amp_type(model)
    local_rank = config.local_rank
    total_rank = getattr(config, 'total_rank', 0)
    main = local_rank == 0 and total_rank == 0
    if main:
        logger.info(f'use_amp: {config.use_amp}, amp_type: {amp_type}!')
    iters = len(train_loader.dataset) // config.batch_size
    iter_index = 1
    acc_steps = config.accumulation_steps
    assert acc_steps >= 1, 'illegal accumulation_steps!'
    lag = getattr(config, 'host_sync_lag', 2)
    scaler = getattr(config, 'scaler', None) if config.use_amp else None
    clip_value = getattr(config, 'clip_grad_value', 0) or 0
    clip_norm = getattr(config, 'clip_max_norm', 0) or 0
    pending = collections.deque()      # (packed [skip, loss] device tensor, batch size, log line or None)
    carried_bad = None                 # skip flag of earlier micro-steps of this accumulation window

    def drain(keep):
        nonlocal iter_index
        while len(pending) > keep:
            packed, n, log_fmt = pending.popleft()
            skip, loss_sum = packed.tolist()
            if skip:
                if main:
                    logger.info('skip this batch!')
                iter_index -= 1            # the reference does not count a skipped iteration
                continue
            loss = loss_sum / float(config.gpus_num)
            losses.update(loss, n)
            if log_fmt is not None and main:
                logger.info(log_fmt.format(loss=loss * acc_steps))

    def forward_backward(images, labels, boundary):
        """forward, loss, (scaled) backward; -> packed [skip flag, loss / acc_steps] of this rank"""
        # device-side replacement of the reference's isinf/isnan python branches (:147-151)
        bad = any_nonfinite(images, labels) if labels.dtype.is_floating_point else any_nonfinite(images)
        if config.use_amp:
            with autocast(device_type=device.type, dtype=amp_type):
                outputs = model(images)
                loss = criterion(outputs, labels)
        else:
            outputs = model(images)
            loss = criterion(outputs, labels)
        bad = bad | (loss == 0.) | ~torch.isfinite(loss)
        loss = loss / acc_steps
        scaled = scaler.scale(loss) if scaler is not None else loss
        if boundary:
            scaled.backward()
        else:
            with model.no_sync():       # no gradient exchange on non-boundary micro-steps
                scaled.backward()
        # one tiny all-reduce carries the skip flag (any rank) and the loss (sum over ranks)
        packed = torch.stack([bad.float(), loss.detach().float()])
        all_reduce_sum_packed(packed, model, config.group)
        return packed

    def update(packed):
        """gradient sync wait, inf/nan check, unscale + clip, fused optimizer step, scaler update, zero_grad, EMA"""
        if hasattr(model, 'finish_gradient_sync'):
            model.finish_gradient_sync()
        skip_flag = packed[0:1]
        if getattr(config, 'skip_inf_nan_grad', False) or scaler is not None:
            optimizer.check_finite()
            skip_flag = torch.maximum(skip_flag, optimizer.found_inf)
        inv_scale = scaler.state[2:3] if scaler is not None else None
        if clip_value > 0:             # reference :211-218: unscale, clamp every gradient element, then the norm clip
            optimizer.clip_grad_value_(clip_value, inv_scale)
            inv_scale = None
        if clip_norm > 0:
            optimizer.clip_grad_norm_(clip_norm, inv_scale)
            inv_scale = None
        optimizer.step(inv_scale, skip_flag)
        if scaler is not None:
            scaler._found_inf = optimizer.found_inf
            scaler.update()
        optimizer.zero_grad()
        if config.use_ema_model:
            config.ema_model.update(model, skip_flag)

    # config.use_step_graph: the whole iteration (forward .. zero_grad) is captured once into a hipGraph and
    # replayed (engine.StepGraph) -- the host then issues one graph launch instead of ~700 kernel launches.
    # Needs accumulation_steps == 1 and static shapes (drop_last loaders); everything the host changes per
    # iteration (the scheduler's learning rates) reaches the captured kernels through the optimizer's device table.
    step_graph = None
    if getattr(config, 'use_step_graph', False) and acc_steps == 1 and device.type == 'cuda':
        from .. import engine

        def whole_step(images, labels):
            packed = forward_backward(images, labels, True)
            update(packed)
            return packed
        cache = getattr(config, '_saicv_step_graphs', None)
        if cache is None:
            cache = {}
            config._saicv_step_graphs = cache       # config is usually a class: the graph lives across epochs
        key = (id(model), id(optimizer))
        step_graph = cache.get(key)
        if step_graph is None:
            step_graph = engine.StepGraph(whole_step, warmup=getattr(config, 'step_graph_warmup', 3),
                                          before_replay=(optimizer.refresh_hyper,))
            cache[key] = step_graph

    micro = 0      # accumulation phase counts micro-batches as they are issued; the lagged skip correction below only
                   # moves the logged / scheduled iteration index, never the phase of a window already under way
    for data in train_loader:
        images, labels = data['image'].to(device, non_blocking=True), data['label'].to(device, non_blocking=True)
        micro += 1
        boundary = micro % acc_steps == 0
        if step_graph is not None:
            packed = step_graph(images, labels).clone()     # the static output is overwritten by the next replay
        else:
            packed = forward_backward(images, labels, boundary)
        if carried_bad is not None:
            packed = torch.stack([torch.maximum(packed[0], carried_bad), packed[1]])
        carried_bad = None if boundary else packed[0]

        if boundary:
            if step_graph is None:
                update(packed)
            scheduler.step(optimizer, iter_index / iters + (epoch - 1))
            log_fmt = None
            if iter_index % int(config.print_interval * acc_steps) == 0:
                log_fmt = (f'train: epoch {epoch:0>4d}, iter [{int(iter_index // acc_steps):0>5d}, '
                           f'{int(iters // acc_steps):0>5d}], lr: {scheduler.current_lr:.6f}, ' + 'loss: {loss:.4f}')
            pending.append((packed, images.size(0), log_fmt))
        drain(lag)
        iter_index += 1
    drain(0)
    return losses.avg * acc_steps


def all_ranks_agree(mine, config):
    """True iff `mine` is true on EVERY rank of config.group -- a host-side decision made the same way everywhere.  The DETR loop
    leaves the captured step for a batch whose images carry more boxes than config.max_annots (graph_inputs -> None); under DDP one
    rank running the eager step (host-side assignment, eagerly issued bucket all-reduces) while the others replay the graph (captured
    collectives) is two different collective sequences: a hang.  The flag travels over a gloo group (created once per
    config, by every rank at the same iteration): no device synchronisation, ~0.1 ms of host time against a 23 ms step."""
    world = int(getattr(config, 'gpus_num', 1) or 1)
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return bool(mine)
    group = getattr(config, '_saicv_host_group', None)
    if group is None:
        own = getattr(config, 'group', None)
        group = own if dist.get_backend(own) == 'gloo' else dist.new_group(backend='gloo')
        config._saicv_host_group = group
    flag = torch.tensor([1 if mine else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group if group is not None else dist.group.WORLD)
    return bool(int(flag[0]))


def _epoch_loop(train_loader, model, optimizer, scheduler, epoch, logger, config, step_fn, total_name, iter_width,
                graph_inputs=None, log_terms=True):
    """Shared iteration engine of the dict-loss loops (detection here; the SAM loop in
    interactive_segmentation_scripts.py follows the same scheme): `step_fn(data)` runs forward + loss and
    returns (bad flag tensor, {name: loss tensor}, batch size).  Everything else -- accumulation, the single
    packed all-reduce of [skip, total, terms...], device-side skip, clipping, scaler, EMA, scheduler, lagged
    host reads and the reference log line -- is common.

    config.use_step_graph + `graph_inputs(data) -> tuple of device tensors` (r04): the WHOLE iteration -- forward, criterion,
    backward with its bucket all-reduces, unscale / clip / fused optimizer step, zero_grad, EMA -- is captured once
    (engine.StepGraph) and replayed, exactly as train_classification does; `step_fn` is then called with that tuple.  Only for
    criteria without host reads and with static shapes (RetinaLoss with SmoothL1; DETR's Hungarian assignment and the
    positive-only IoU branches of FCOSLoss are not), accumulation_steps == 1 and fixed-size batches."""
    losses = AverageMeter()
    local_rank = config.local_rank
    main = local_rank == 0 and getattr(config, 'total_rank', 0) == 0
    iters = len(train_loader.dataset) // config.batch_size
    iter_index = 1
    acc_steps = config.accumulation_steps
    assert acc_steps >= 1, 'illegal accumulation_steps!'
    lag = getattr(config, 'host_sync_lag', 2)
    scaler = getattr(config, 'scaler', None) if config.use_amp else None
    clip_norm = getattr(config, 'clip_max_norm', 0) or 0
    clip_value = getattr(config, 'clip_grad_value', 0) or 0
    pending = collections.deque()
    carried_bad = None
    # names of the criterion's loss terms, learnt on the first forward.  A cached step graph (below) replays WITHOUT running
    # forward_backward again, so the names live in a holder that is cached with the graph: a second epoch in the same process reuses
    # both (a per-call `keys = None` stayed None from epoch 2 on and the first logged iteration raised on rank 0)
    names = {'keys': None}

    def drain(keep):
        nonlocal iter_index
        while len(pending) > keep:
            packed, n, log_fmt = pending.popleft()
            vals = packed.tolist()
            if vals[0]:
                if main:
                    logger.info('skip this batch!')
                iter_index -= 1
                continue
            loss = vals[1] / float(config.gpus_num)
            losses.update(loss, n)
            if log_fmt is not None and main:
                terms = ''.join(f'{k}: {v / float(config.gpus_num) * acc_steps:.4f}, ' for k, v in zip(names['keys'], vals[2:]))
                logger.info(log_fmt.format(loss=loss * acc_steps) + (terms if log_terms else ''))

    def forward_backward(data, boundary):
        """forward, criterion, (scaled) backward; -> (packed [skip, total, terms...] reduced over the ranks, batch size)"""
        bad, loss_value, n = step_fn(data)
        return loss_tail(bad, loss_value, boundary), n

    def loss_tail(bad, loss_value, boundary):
        """loss terms -> total, skip flag, (scaled) backward, the packed vector reduced over the ranks"""
        if names['keys'] is None:
            names['keys'] = list(loss_value.keys())
        keys = names['keys']
        if len(keys) > 4:
            # many terms (DETR: 18): one stack + one sum instead of a chain of adds and a second stack; d loss / d term is 1 either way
            stacked = torch.stack([loss_value[k].float() for k in keys])
            loss = stacked.sum()
            terms = stacked.detach() / acc_steps
        else:
            loss = sum(loss_value.values())
            terms = torch.stack([loss_value[k].detach().float() for k in keys]) / acc_steps
        bad = bad | (loss == 0.) | ~torch.isfinite(loss) | ~torch.isfinite(terms).all()
        loss = loss / acc_steps
        scaled = scaler.scale(loss) if scaler is not None else loss
        if boundary:
            scaled.backward()
        else:
            with model.no_sync():
                scaled.backward()
        packed = torch.cat([torch.stack([bad.float(), loss.detach().float()]), terms])
        all_reduce_sum_packed(packed, model, config.group)
        return packed

    def update(packed):
        if hasattr(model, 'finish_gradient_sync'):
            model.finish_gradient_sync()
        skip_flag = packed[0:1]
        if getattr(config, 'skip_inf_nan_grad', False) or scaler is not None:
            optimizer.check_finite()
            skip_flag = torch.maximum(skip_flag, optimizer.found_inf)
        inv_scale = scaler.state[2:3] if scaler is not None else None
        if clip_value > 0:
            optimizer.clip_grad_value_(clip_value, inv_scale)
            inv_scale = None
        if clip_norm > 0:
            optimizer.clip_grad_norm_(clip_norm, inv_scale)
            inv_scale = None
        optimizer.step(inv_scale, skip_flag)
        if scaler is not None:
            scaler._found_inf = optimizer.found_inf
            scaler.update()
        optimizer.zero_grad()
        if getattr(config, 'use_ema_model', False):
            config.ema_model.update(model, skip_flag)

    step_graph = None
    if (graph_inputs is not None and getattr(config, 'use_step_graph', False) and acc_steps == 1
            and _device_of(model).type == 'cuda'):
        from .. import engine
        batch_n = [0]

        def whole_step(*tensors):
            packed, n = forward_backward(tensors, True)
            batch_n[0] = n
            update(packed)
            return packed

        cache = getattr(config, '_saicv_step_graphs', None)
        if cache is None:
            cache = {}
            config._saicv_step_graphs = cache
        key = (id(model), id(optimizer))
        step_graph = cache.get(key)
        if step_graph is None:
            step_graph = engine.StepGraph(whole_step, warmup=getattr(config, 'step_graph_warmup', 3),
                                          before_replay=(optimizer.refresh_hyper,))
            step_graph.loss_term_names = names      # the captured closure writes into THIS holder
            cache[key] = step_graph
        else:
            names = step_graph.loss_term_names

    micro = 0      # accumulation phase by issued micro-batch (see train_classification)
    for data in train_loader:
        micro += 1
        boundary = micro % acc_steps == 0
        tensors = graph_inputs(data) if step_graph is not None else None     # None: this batch does not fit the captured shapes
        if step_graph is not None and getattr(graph_inputs, 'may_decline', False) and not all_ranks_agree(tensors is not None, config):
            tensors = None                                                   # ... on SOME rank: every rank takes the eager step
        if tensors is not None:
            packed = step_graph(*tensors).clone()
            n = tensors[0].size(0)
        else:
            packed, n = forward_backward(data, boundary)
        if carried_bad is not None:
            packed = torch.cat([torch.maximum(packed[0:1], carried_bad), packed[1:]])
        carried_bad = None if boundary else packed[0:1]
        if boundary:
            if tensors is None:
                update(packed)
            scheduler.step(optimizer, iter_index / iters + (epoch - 1))
            log_fmt = None
            if iter_index % int(config.print_interval * acc_steps) == 0:
                log_fmt = (f'train: epoch {epoch:0>4d}, iter [{int(iter_index // acc_steps):0>{iter_width}d}, '
                           f'{int(iters // acc_steps):0>{iter_width}d}], lr: {scheduler.current_lr:.6f}, ' +
                           total_name + (': {loss:.4f}, ' if log_terms else ': {loss:.4f}'))
            pending.append((packed, n, log_fmt))
        drain(lag)
        iter_index += 1
    drain(0)
    return losses.avg * acc_steps


def train_detection(train_loader, model, criterion, optimizer, scheduler, epoch, logger, config):
    '''train detection model for one epoch (reference tools/scripts.py:900-1092).  DETR-family networks
    ('detr' in config.network) take the padding mask and the normalised cxcywh annotations.'''
    model.train()
    device = _device_of(model)
    amp_type = get_amp_type(model)
    if config.local_rank == 0 and getattr(config, 'total_rank', 0) == 0:
        logger.info(f'use_amp: {config.use_amp}, amp_type: {amp_type}!')
    is_detr = 'detr' in config.network

    static_detr = is_detr and getattr(criterion, 'static_form', False) and getattr(config, 'use_step_graph', False)

    def step_fn(data):
        if isinstance(data, tuple) and static_detr:
            # captured DETR step (r05): fixed shapes and no host read -- the cost matrices, the Hungarian assignment (saicv_detr_assign)
            # and the loss over the padded ground truth all stay on the device
            images, mask, ann = data
            bad = any_nonfinite(images, ann)
            with autocast(device_type=device.type, dtype=amp_type, enabled=bool(config.use_amp)):
                outs = model(images, mask)
                cost, valid = criterion.match_inputs(outs, ann)
                src, tgt, w = criterion.assign_device(cost, valid)
                loss_value = criterion.forward_static(outs, ann, src, tgt, w)
            return bad, loss_value, images.size(0)
        if isinstance(data, tuple):                      # captured step: (images, targets) already on the device, static buffers
            images, targets = data
        else:
            images = data['image'].to(device, non_blocking=True)
            host_targets = data['scaled_annots'] if is_detr else data['annots']
            targets = host_targets.to(device, non_blocking=True)
            if is_detr and not host_targets.is_cuda:
                targets._saicv_host = host_targets       # DETRLoss selects the valid rows on the host (no per-image device sync)
        bad = any_nonfinite(images, targets)
        with autocast(device_type=device.type, dtype=amp_type, enabled=bool(config.use_amp)):
            if not is_detr:
                outs = model(images)
            elif getattr(config, 'device_pad_mask', False):      # f3: the mask from the [B, 2] sizes instead of a [B, S, S] copy
                outs = model(images, pad_mask_on_device(data['scaled_size'], images.shape[-1], device))
            else:
                outs = model(images, data['mask'].to(device, non_blocking=True))
            loss_value = criterion(outs, targets)
        return bad, loss_value, images.size(0)

    # the dense detectors' step has no host read (anchor assignment, focal loss and SmoothL1 are decided on the device): it can be
    # captured whole.  So can DETR's since r05: its Hungarian assignment runs on the device (DETRLoss.assign_device) over the
    # annotations padded (class -1 rows) to config.max_annots rows (default 100 = the reference's query count; a batch with more boxes
    # in one image takes the eager step with the host-side assignment).  Criteria that index by a data-dependent positive mask
    # (`capturable = False`, e.g. the IoU branches) have dynamic shapes and stay eager.
    graph_inputs = None
    if not is_detr and getattr(criterion, 'capturable', False):
        def graph_inputs(data):
            return (data['image'].to(device, non_blocking=True), data['annots'].to(device, non_blocking=True))
    elif static_detr:
        max_annots = int(getattr(config, 'max_annots', 100))

        def graph_inputs(data):
            ann = data['scaled_annots']
            if ann.shape[1] > max_annots:
                if bool((ann[:, max_annots:, 4] >= 0).any()):
                    return None
                ann = ann[:, :max_annots]
            ann = ann.to(device, non_blocking=True).float()
            if ann.shape[1] < max_annots:
                ann = torch.cat([ann, ann.new_full((ann.shape[0], max_annots - ann.shape[1], ann.shape[2]), -1.0)], dim=1)
            if getattr(config, 'device_pad_mask', False):
                mask = pad_mask_on_device(data['scaled_size'], data['image'].shape[-1], device)
            else:
                mask = data['mask'].to(device, non_blocking=True)
            return (data['image'].to(device, non_blocking=True), mask, ann)
        graph_inputs.may_decline = True
    return _epoch_loop(train_loader, model, optimizer, scheduler, epoch, logger, config, step_fn, 'total_loss', 5, graph_inputs)


def train_mae_self_supervised_learning(train_loader, model, criterion, optimizer, scheduler, epoch, logger, config):
    '''train mae self supervised model for one epoch (reference tools/scripts.py:1774-1934): `outputs, masks = model(images)`,
    `loss = criterion(outputs, labels, masks)` (the per-patch regression target comes from MAESelfSupervisedPretrainCollater),
    the reference's skip / accumulation / clipping / scaler / EMA / scheduler semantics and its log line
    `train: epoch 0001, iter [00100, 01251], lr: 0.000150, loss: 0.7523`.  The iteration has no host read and static shapes
    (the mask is an argsort of device noise, the kept-patch count is fixed by mask_ratio): with config.use_step_graph it is
    captured whole and replayed, as the classification loop is.'''
    model.train()
    device = _device_of(model)
    amp_type = get_amp_type(model)
    if config.local_rank == 0 and getattr(config, 'total_rank', 0) == 0:
        logger.info(f'use_amp: {config.use_amp}, amp_type: {amp_type}!')

    def step_fn(data):
        if isinstance(data, tuple):                      # captured step: static device buffers
            images, labels = data
        else:
            images = data['image'].to(device, non_blocking=True)
            labels = data['label'].to(device, non_blocking=True)
        bad = any_nonfinite(images, labels)
        with autocast(device_type=device.type, dtype=amp_type, enabled=bool(config.use_amp)):
            outputs, masks = model(images)
            loss = criterion(outputs, labels, masks)
        return bad, {'loss': loss}, images.size(0)

    def graph_inputs(data):
        return (data['image'].to(device, non_blocking=True), data['label'].to(device, non_blocking=True))
    return _epoch_loop(train_loader, model, optimizer, scheduler, epoch, logger, config, step_fn, 'loss', 5, graph_inputs,
                       log_terms=False)


# ---------------------------------------------------------------------------------------------- detection evaluation
def compute_voc_ap(recall, precision, use_07_metric=False):
    """area under the precision envelope (VOC >= 2010) or the 11-point average (VOC 2007); reference :503-532"""
    recall, precision = np.asarray(recall, dtype=np.float64), np.asarray(precision, dtype=np.float64)
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            hit = recall >= t
            ap = ap + (np.max(precision[hit]) if hit.any() else 0) / 11.
        return ap
    r = np.concatenate(([0.], recall, [1.]))
    env = np.maximum.accumulate(np.concatenate(([0.], precision, [0.]))[::-1])[::-1]      # precision envelope from the right
    step = np.where(r[1:] != r[:-1])[0]
    return np.sum((r[step + 1] - r[step]) * env[step + 1])


def compute_ious(a, b):
    """[N, 4] x [M, 4] xyxy boxes -> IoU [N, M] (no clamp on degenerate unions, as the reference :535-556)"""
    a, b = np.expand_dims(a, axis=1), np.expand_dims(b, axis=0)
    overlap = np.prod(np.maximum(0.0, np.minimum(a[..., 2:], b[..., 2:]) - np.maximum(a[..., :2], b[..., :2])), axis=-1)
    area_a = np.prod(a[..., 2:] - a[..., :2], axis=-1)
    area_b = np.prod(b[..., 2:] - b[..., :2], axis=-1)
    return overlap / (area_a + area_b - overlap)


def _voc_class_ap(gt_boxes, pred_boxes, pred_scores, threshold):
    """AP of one class at one IoU threshold.  Per image, predictions are visited in the order the decoder emitted them; each
    takes its best-IoU ground-truth box if that box is still free and the IoU reaches the threshold (reference :676-706)."""
    flags, scores, total = [], [], 0
    for boxes, preds, sc in zip(gt_boxes, pred_boxes, pred_scores):
        total += len(boxes)
        if len(preds) == 0:
            continue
        scores.append(sc)
        hit = np.zeros(len(preds))
        if boxes.shape[0] > 0:
            iou = compute_ious(boxes, preds)                              # [gt, pred]
            best = np.argmax(iou, axis=0)
            best_iou = iou[best, np.arange(len(preds))]
            taken = np.zeros(boxes.shape[0], dtype=bool)
            for j in range(len(preds)):
                if best_iou[j] >= threshold and not taken[best[j]]:
                    hit[j] = 1
                    taken[best[j]] = True
        flags.append(hit)
    if not scores:
        tp = fp = np.zeros((0,))
    else:
        order = np.argsort(-np.concatenate(scores))
        hit = np.concatenate(flags)[order]
        tp, fp = np.cumsum(hit), np.cumsum(1 - hit)
    with np.errstate(divide='ignore', invalid='ignore'):
        recall = tp / total
    precision = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return compute_voc_ap(recall, precision, use_07_metric=False)


def evaluate_voc_detection(test_loader, model, criterion, decoder, config):
    """VOC-style mAP at every threshold of config.eval_voc_iou_threshold_list (reference :559-739): forward + loss + decode per
    batch, boxes back to the original image scale and clipped to it, then per class and threshold the matching above."""
    model.eval()
    batch_time, data_time, losses = AverageMeter(), AverageMeter(), AverageMeter()
    batch_size = int(config.batch_size // config.gpus_num)
    device = _device_of(model)
    on_gpu = device.type == 'cuda'
    is_detr = 'detr' in config.network
    preds, gts = [], []
    with torch.no_grad():
        end = time.time()
        for data in test_loader:
            images, annots, scales, sizes = data['image'].to(device), data['annots'].to(device), data['scale'], data['size']
            if on_gpu:
                torch.cuda.synchronize()
            data_time.update(time.time() - end, images.size(0))
            end = time.time()
            outs = model(images, data['mask'].to(device)) if is_detr else model(images)
            loss = sum(criterion(outs, annots).values())
            losses.update(float(loss), images.size(0))
            pred_scores, pred_classes, pred_boxes = decoder(outs, data['scaled_size']) if is_detr else decoder(outs)
            unscale = np.expand_dims(np.expand_dims(scales, axis=-1), axis=-1)
            pred_boxes = pred_boxes / unscale
            if on_gpu:
                torch.cuda.synchronize()
            batch_time.update(time.time() - end, images.size(0))
            annots = annots.cpu().numpy()
            gt_bboxes, gt_classes = annots[:, :, 0:4] / unscale, annots[:, :, 4]
            for sc, cl, bx, gb, gc, size in zip(pred_scores, pred_classes, pred_boxes, gt_bboxes, gt_classes, sizes):
                live = cl > -1
                sc, cl, bx = sc[live], cl[live], bx[live].copy()
                bx[:, 0:2] = np.maximum(bx[:, 0:2], 0)
                bx[:, 2] = np.minimum(bx[:, 2], size[1])
                bx[:, 3] = np.minimum(bx[:, 3], size[0])
                preds.append([bx, cl, sc])
                gts.append([gb[gc > -1], gc[gc > -1]])
            end = time.time()
    result_dict = collections.OrderedDict()
    result_dict['test_loss'] = losses.avg
    result_dict['per_image_load_time'] = f'{data_time.avg / batch_size * 1000:.3f}ms'
    result_dict['per_image_inference_time'] = f'{batch_time.avg / batch_size * 1000:.3f}ms'
    per_class = collections.OrderedDict()
    for thr in config.eval_voc_iou_threshold_list:
        aps = collections.OrderedDict()
        for c in range(config.num_classes):
            aps[c] = 100 * _voc_class_ap([g[0][g[1] == c] for g in gts], [p[0][p[1] == c] for p in preds],
                                         [p[2][p[1] == c] for p in preds], thr)
        result_dict[f'IoU={thr:.2f},area=all,maxDets=100,mAP'] = sum(float(v) for v in aps.values()) / config.num_classes
        per_class[f'IoU={thr:.2f},area=all,maxDets=100,per_class_ap'] = aps
    result_dict.update(per_class)
    return result_dict


def evaluate_coco_detection(test_loader, model, criterion, decoder, config):
    """COCO-style evaluation (reference :742-881): forward + loss + decode per batch, boxes back to the original image scale,
    clipped, converted to [x, y, w, h], then the twelve COCO summary numbers (x 100) under the reference's keys.  The reference
    hands the detections to pycocotools' COCOeval; here the same protocol runs in numpy (tools/cocoeval_numpy.py).  Ground truth:
    `config.test_dataset.coco.dataset` (a COCO annotation dict, as pycocotools holds it) when the dataset has one, else the
    annotations the loader delivers (un-scaled; every box a regular, non-crowd object with area w * h) -- which is what the
    synthetic benchmark datasets provide."""
    from . import cocoeval_numpy as CE
    model.eval()
    batch_time, data_time, losses = AverageMeter(), AverageMeter(), AverageMeter()
    test_dataset = config.test_dataset
    batch_size = int(config.batch_size // config.gpus_num)
    device = _device_of(model)
    on_gpu = device.type == 'cuda'
    is_detr = 'detr' in config.network
    image_id_of = getattr(test_dataset, 'image_ids', None)
    cat_of = getattr(test_dataset, 'coco_label_to_cat_id', None)
    coco = getattr(test_dataset, 'coco', None)
    results, gts, image_ids = [], [], []
    with torch.no_grad():
        end = time.time()
        for i, data in enumerate(test_loader):
            images, annots, scales, sizes = data['image'].to(device), data['annots'].to(device), data['scale'], data['size']
            if on_gpu:
                torch.cuda.synchronize()
            data_time.update(time.time() - end, images.size(0))
            end = time.time()
            outs = model(images, data['mask'].to(device)) if is_detr else model(images)
            loss = sum(criterion(outs, annots).values())
            losses.update(float(loss), images.size(0))
            scores, classes, boxes = decoder(outs, data['scaled_size']) if is_detr else decoder(outs)
            unscale = np.expand_dims(np.expand_dims(scales, axis=-1), axis=-1)
            boxes = boxes / unscale
            if on_gpu:
                torch.cuda.synchronize()
            batch_time.update(time.time() - end, images.size(0))
            annots_np = annots.cpu().numpy()
            for b in range(images.size(0)):
                index = i * batch_size + b
                img_id = image_id_of[index] if image_id_of is not None else index
                image_ids.append(img_id)
                pb = boxes[b].copy()
                pb[:, 0], pb[:, 1] = np.maximum(pb[:, 0], 0), np.maximum(pb[:, 1], 0)
                pb[:, 2], pb[:, 3] = np.minimum(pb[:, 2], sizes[b][1]), np.minimum(pb[:, 3], sizes[b][0])
                pb[:, 2:] -= pb[:, :2]                                  # COCO boxes are [x_min, y_min, w, h]
                for sc, cl, bx in zip(scores[b], classes[b], pb):
                    if int(cl) == -1:
                        break
                    results.append({'image_id': img_id, 'category_id': cat_of[int(cl)] if cat_of is not None else int(cl),
                                    'score': float(sc), 'bbox': bx.tolist()})
                if coco is None:
                    for row in annots_np[b]:
                        if row[4] < 0:
                            continue
                        x0, y0, x1, y1 = (row[0:4] / float(np.asarray(scales[b]).reshape(-1)[0])).tolist()
                        gts.append({'image_id': img_id, 'category_id': cat_of[int(row[4])] if cat_of is not None else int(row[4]),
                                    'bbox': [x0, y0, x1 - x0, y1 - y0], 'area': (x1 - x0) * (y1 - y0), 'iscrowd': 0})
            end = time.time()
    result_dict = collections.OrderedDict()
    result_dict['test_loss'] = losses.avg
    result_dict['per_image_load_time'] = f'{data_time.avg / batch_size * 1000:.3f}ms'
    result_dict['per_image_inference_time'] = f'{batch_time.avg / batch_size * 1000:.3f}ms'
    if len(results) == 0:
        for name in CE.STAT_NAMES:
            result_dict[name] = 0
        return result_dict
    cats = None
    if coco is not None:
        ds = coco.dataset
        wanted = set(image_ids)
        gts = [dict(a) for a in ds['annotations'] if a['image_id'] in wanted]
        cats = [c['id'] for c in ds['categories']]
    stats, _, _ = CE.evaluate_bbox(gts, results, image_ids=image_ids, category_ids=cats)
    for name, v in zip(CE.STAT_NAMES, stats):
        result_dict[name] = v * 100
    return result_dict


def test_detection(test_loader, model, criterion, decoder, config):
    """reference :884-897"""
    assert config.eval_type in ['COCO', 'VOC']
    if getattr(config, 'use_ema_model', False):
        model = config.ema_model.ema_model
    return {'COCO': evaluate_coco_detection, 'VOC': evaluate_voc_detection}[config.eval_type](test_loader, model, criterion, decoder, config)
