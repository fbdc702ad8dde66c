"""Per-iteration SAM training loop of the reference tools/interactive_segmentation_scripts.py on the
MI355X engine.

  sample_error_click                                  (reference sample_random_point, :202-228; one HIP kernel)
  get_decoder_iters_prompt_points_and_prompt_mask     (reference :231-271)
  train_sam_segmentation                              (reference :274-533)

Loop semantics are kept: one image-encoder forward, 1 + decoder_iters prompt-encoder / mask-decoder
passes whose new prompts (an error-region point and the best mask at 1/4 resolution) are sampled
without gradient from the previous pass, SAMLoss over all passes, gradient accumulation with
`no_sync()`, gradient-norm clipping, GradScaler, per-iteration scheduler, the reference log line with
the three loss terms.  Differences, as in tools/scripts.py of this package: the manual per-parameter
all-reduce loop (:446-449, one collective per tensor) is the engine's bucketed all-reduce, already
overlapped with backward; skip decisions stay on the device and reach the host `host_sync_lag`
iterations later; no per-iteration barrier.
"""
import collections
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.amp.autocast_mode import autocast

from ..engine import any_nonfinite
from ..SimpleAICV.classification.common import AverageMeter, get_amp_type
from .scripts import _device_of, _dist_on, all_reduce_sum_packed


_click_serial = [0]
_click_seed_dev = [None]      # int32[1] on the device while a step that may be captured runs: the kernel adds it to the baked-in seed


def sample_error_click(gt_masks, mask_logits=None, channel=None, gt_threshold=0.5, pred_threshold=0.0, seed=None):
    """[B, 1, 3] (x, y, label): one click per sample drawn uniformly from the error region of the current prediction --
    label 1 on a missed foreground pixel, label 0 on a falsely predicted one, and on a background pixel when the
    prediction is already exact (what reference tools/interactive_segmentation_scripts.py:202-228 samples through an
    arg-max over a [B, 1, H, W, 2] noise tensor).  Here: saicv_sam_sample_point -- one pass over the masks, a counter-based
    draw per (pixel, label) slot, no noise tensor, no gather of the best mask (the kernel reads channel `channel[b]` of the
    [B, M, H, W] logits in place).  gt_masks: [B, 1, H, W] fp32; mask_logits: [B, M, H, W] bf16 / fp32 or None."""
    from .._lib import check, dtype_code, lib, ptr, require_gpu, stream
    require_gpu(gt_masks)
    b, _, h, w = gt_masks.shape
    gt = gt_masks.float().contiguous()
    if mask_logits is not None:
        mask_logits = mask_logits.contiguous()
        if mask_logits.dtype not in (torch.float32, torch.bfloat16):
            mask_logits = mask_logits.float()
        assert mask_logits.shape[0] == b and tuple(mask_logits.shape[-2:]) == (h, w)
        if channel is not None:
            channel = channel.to(torch.int64).contiguous()
    if seed is None:
        _click_serial[0] += 1
        seed = (torch.initial_seed() * 2654435761 + _click_serial[0]) & 0xffffffff
    keys = torch.empty(b * 4, dtype=torch.int64, device=gt.device)
    points = torch.empty((b, 1, 3), dtype=torch.float32, device=gt.device)
    m = mask_logits.shape[1] if mask_logits is not None else 1
    # inside a step that is (or will be) captured, `seed` is frozen with the graph: the varying part comes from device memory
    check(lib().saicv_sam_sample_point_dseed(dtype_code(mask_logits.dtype) if mask_logits is not None else 1, ptr(gt), ptr(mask_logits),
                                             h * w, ptr(channel), m, float(gt_threshold), float(pred_threshold), int(seed),
                                             ptr(_click_seed_dev[0]), ptr(keys), ptr(points), b, h, w, stream()), 'sam_sample_point')
    return points


def get_decoder_iters_prompt_points_and_prompt_mask(mask_preds, iou_preds, gt_masks, prompts, config):
    """Prompts of the next decoder pass (reference :231-271): the click above against the best-IoU mask of this pass, appended
    to the point prompts, and that mask at 1/4 resolution as the mask prompt."""
    with torch.no_grad():
        if mask_preds.dim() == 5:
            mask_preds = mask_preds.squeeze(2)
        n_out = iou_preds.shape[1]
        best = torch.argmax(iou_preds, dim=-1) if n_out > 1 else None
        click = sample_error_click(gt_masks, mask_preds, best, 0.5, config.mask_threshold)
        if best is not None:
            best_mask = mask_preds[torch.arange(mask_preds.shape[0], device=mask_preds.device), best].unsqueeze(1)
        else:
            best_mask = mask_preds
        old = prompts['prompt_point']
        prompts['prompt_point'] = click if old is None else torch.cat([old, click], dim=1)
        q = config.input_image_size // 4
        prompts['prompt_mask'] = F.interpolate(best_mask.float(), size=(q, q), mode='bilinear')
    return prompts


def _choose_prompts(config, prompt_points, prompt_boxs, prompt_masks, device):
    """Prompt-type draw of the reference loop (:314-353) -> (prompts, decoder_iters)."""
    prompts = {'prompt_point': None, 'prompt_box': None, 'prompt_mask': None}
    p_point, p_box, p_mask = (config.prompt_probs['prompt_point'], config.prompt_probs['prompt_box'],
                              config.prompt_probs['prompt_mask'])
    assert 0.0 <= p_point <= 1.0 and 0.0 <= p_box <= 1.0 and 0.0 <= p_mask <= 1.0
    decoder_iters = config.decoder_iters
    if config.use_single_prompt:
        assert sum(config.prompt_probs.values()) == 1.
        u = np.random.uniform(0, 1)
        if 0. < u < p_point:
            prompts['prompt_point'] = prompt_points.to(device)
        elif p_point < u < (p_point + p_box):
            prompts['prompt_box'] = prompt_boxs.to(device)
        elif (p_point + p_box) < u < 1.:
            prompts['prompt_mask'] = prompt_masks.to(device)
            decoder_iters = 0
    else:
        assert sum(config.prompt_probs.values()) <= 3.
        u_point, u_box, u_mask = np.random.uniform(0, 1), np.random.uniform(0, 1), np.random.uniform(0, 1)
        if u_point < p_point:
            prompts['prompt_point'] = prompt_points.to(device)
        if u_box < p_box:
            prompts['prompt_box'] = prompt_boxs.to(device)
        if prompts['prompt_point'] is None and prompts['prompt_box'] is None:
            prompts['prompt_point'] = prompt_points.to(device)
            prompts['prompt_box'] = prompt_boxs.to(device)
        if u_mask < p_mask:
            prompts['prompt_mask'] = prompt_masks.to(device)
            decoder_iters = 0
    return prompts, decoder_iters


def train_sam_segmentation(train_loader, model, criterion, optimizer, scheduler, epoch, logger, config):
    losses = AverageMeter()
    model.train()
    if config.frozen_image_encoder:
        model.module.image_encoder.eval()
    if config.frozen_prompt_encoder:
        model.module.prompt_encoder.eval()
    if config.frozen_mask_decoder:
        model.module.mask_decoder.eval()
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
    clip_norm = getattr(config, 'clip_max_norm', 0) or 0
    clip_value = getattr(config, 'clip_grad_value', 0) or 0
    pending = collections.deque()
    carried_bad = None
    net = model.module
    keys = ('focal_loss', 'dice_loss', 'iou_predict_loss')

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
                terms = ''.join(f'{k}: {v / float(config.gpus_num) * acc_steps:.4f}, ' for k, v in zip(keys, vals[2:]))
                logger.info(log_fmt.format(loss=loss * acc_steps) + terms)

    def amp():
        return autocast(device_type=device.type, dtype=amp_type, enabled=bool(config.use_amp))

    def forward_backward(images, masks, prompts, decoder_iters, boundary):
        """image encoder, 1 + decoder_iters prompt / decoder passes, SAMLoss, (scaled) backward -> packed [skip, loss / acc_steps, terms]"""
        bad = any_nonfinite(images)
        with amp():
            batch_image_embeddings = net.forward_image_encoder(images)
            mask_preds, iou_preds = net.forward_prompt_encoder_mask_decoder(batch_image_embeddings, prompts,
                                                                            mask_out_idxs=config.mask_out_idxs)
        all_iter_mask_preds, all_iter_iou_preds = [mask_preds], [iou_preds]
        for _ in range(decoder_iters):
            prompts = get_decoder_iters_prompt_points_and_prompt_mask(mask_preds, iou_preds, masks, prompts, config)
            with amp():
                mask_preds, iou_preds = net.forward_prompt_encoder_mask_decoder(batch_image_embeddings, prompts,
                                                                                mask_out_idxs=config.mask_out_idxs)
            all_iter_mask_preds.append(mask_preds)
            all_iter_iou_preds.append(iou_preds)
        with amp():
            loss_value = criterion([all_iter_mask_preds, all_iter_iou_preds], masks)
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

    # config.use_step_graph (r06): the whole iteration as ONE hipGraph, as in tools/scripts.py -- the step has no host read (the skip
    # decision, the clicks and the best-mask selection are device work).  The prompt-type draw stays on the host, once per iteration
    # BEFORE the step (reference :314-353): it decides which prompt tensors exist and how many decoder passes run, i.e. the SHAPE of
    # the step, so there is one captured graph per drawn combination (at most five), each captured after its own eager warm-up.
    # The click sampler's seed is frozen with a graph; its varying part lives in device memory and is bumped before every step.
    # On ONE GPU the captured step is no faster than eager launches (65.2 vs 65.6 ms at b8: the step is GPU-bound) and the reference
    # config leaves it OFF; what it is for is the bucketed all-reduce on the communication stream under backward, which only a
    # captured step gets (engine.py).
    use_graph = bool(getattr(config, 'use_step_graph', False)) and acc_steps == 1 and device.type == 'cuda'
    # (ROCm's graph packet capture breaks THIS step's graph -- package __init__.py, DESIGN.md section 3k: the package switches it off at
    # import, and engine.StepGraph runs the step eagerly in a process where that came too late)
    graphs = None
    if use_graph:
        from .. import engine
        graphs = getattr(config, '_saicv_step_graphs', None)
        if graphs is None:
            graphs = {}
            config._saicv_step_graphs = graphs
        if _click_seed_dev[0] is None or _click_seed_dev[0].device != device:
            _click_seed_dev[0] = torch.zeros(1, dtype=torch.int32, device=device)

    def graph_for(prompts, decoder_iters):
        names = tuple(k for k in ('prompt_point', 'prompt_box', 'prompt_mask') if prompts[k] is not None)
        key = (id(model), id(optimizer), names, decoder_iters)
        g = graphs.get(key)
        if g is None:
            def whole_step(images, masks, *tensors):
                pr = {'prompt_point': None, 'prompt_box': None, 'prompt_mask': None}
                pr.update(zip(names, tensors))
                packed = forward_backward(images, masks, pr, decoder_iters, True)
                update(packed)
                return packed
            g = engine.StepGraph(whole_step, warmup=getattr(config, 'step_graph_warmup', 3), before_replay=(optimizer.refresh_hyper,))
            graphs[key] = g
        return g, [prompts[k] for k in names]

    micro = 0      # accumulation phase by issued micro-batch (see tools/scripts.py train_classification)
    for data in train_loader:
        micro += 1
        images, masks = data['image'].to(device, non_blocking=True), data['mask'].to(device, non_blocking=True)
        prompts, decoder_iters = _choose_prompts(config, data['prompt_point'], data['prompt_box'], data['prompt_mask'],
                                                 device)
        boundary = micro % acc_steps == 0
        if use_graph:
            _click_seed_dev[0].add_(7919)          # outside the graph: every replay samples its clicks with another seed
            g, tensors = graph_for(prompts, decoder_iters)
            packed = g(images, masks, *tensors).clone()
        else:
            packed = forward_backward(images, masks, prompts, decoder_iters, boundary)
        if carried_bad is not None:
            packed = torch.cat([torch.maximum(packed[0:1], carried_bad), packed[1:]])
        carried_bad = None if boundary else packed[0:1]

        if boundary:
            if not use_graph:
                update(packed)
            scheduler.step(optimizer, iter_index / iters + (epoch - 1))
            log_fmt = None
            if iter_index % int(config.print_interval * acc_steps) == 0:
                log_fmt = (f'train: epoch {epoch:0>4d}, iter [{int(iter_index // acc_steps):0>6d}, '
                           f'{int(iters // acc_steps):0>6d}], lr: {scheduler.current_lr:.6f}, ' + 'loss: {loss:.4f}, ')
            pending.append((packed, images.size(0), log_fmt))
        drain(lag)
        iter_index += 1
    drain(0)
    return losses.avg * acc_steps
