"""Per-iteration SAM training loop of the reference tools/interactive_segmentation_scripts.py on the
MI355X engine.

  sample_random_point                                 (reference :202-228)
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

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.amp.autocast_mode import autocast

from ..SimpleAICV.classification.common import AverageMeter, get_amp_type
from .scripts import _device_of, _dist_on, all_reduce_sum_packed


def sample_random_point(gt_masks, pred_masks, num_pt=1):
    """One click per sample in the error region: label 1 on a false negative, 0 on a false positive
    (or on background when the prediction is already exact)."""
    gt_masks = gt_masks.bool()
    if pred_masks is None:
        pred_masks = torch.zeros_like(gt_masks)
    pred_masks = pred_masks.bool()
    B, _, H_im, W_im = gt_masks.shape
    device = gt_masks.device
    fp_masks = ~gt_masks & pred_masks
    fn_masks = gt_masks & ~pred_masks
    all_correct = torch.all((gt_masks == pred_masks).flatten(2), dim=2)[..., None, None]
    pts_noise = torch.rand(B, num_pt, H_im, W_im, 2, device=device)
    pts_noise[..., 0] *= fp_masks | (all_correct & ~gt_masks)
    pts_noise[..., 1] *= fn_masks
    pts_idx = pts_noise.flatten(2).argmax(dim=2)
    labels = (pts_idx % 2).to(torch.int32)
    pts_idx = pts_idx // 2
    pts_x = pts_idx % W_im
    pts_y = pts_idx // W_im
    points = torch.stack([pts_x, pts_y], dim=2).float()
    return torch.cat([points, labels.unsqueeze(dim=-1)], dim=-1)


def get_decoder_iters_prompt_points_and_prompt_mask(mask_preds, iou_preds, gt_masks, prompts, config):
    with torch.no_grad():
        if len(mask_preds.shape) == 5:
            mask_preds = torch.squeeze(mask_preds, dim=2)
        batch_size, mask_out_idx_num = iou_preds.shape[0], iou_preds.shape[1]
        device = iou_preds.device
        best_iou_masks = mask_preds
        if mask_out_idx_num > 1:
            best_iou_idxs = torch.argmax(iou_preds, dim=-1)
            best_iou_masks = mask_preds[torch.arange(batch_size, device=device), best_iou_idxs].unsqueeze(1)
        new_prompt_points = sample_random_point((gt_masks > 0.5), (best_iou_masks > config.mask_threshold), num_pt=1)
        prompt_points = prompts['prompt_point']
        prompts['prompt_point'] = (torch.cat([prompt_points, new_prompt_points], dim=1)
                                   if prompt_points is not None else new_prompt_points)
        prompts['prompt_mask'] = F.interpolate(best_iou_masks.float(),
                                               size=(config.input_image_size // 4, config.input_image_size // 4),
                                               mode='bilinear')
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
    if (getattr(config, 'clip_grad_value', 0) or 0) > 0:
        raise NotImplementedError('clip_grad_value is not used by the hot-path configs')
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

    micro = 0      # accumulation phase by issued micro-batch (see tools/scripts.py train_classification)
    for data in train_loader:
        micro += 1
        images, masks = data['image'].to(device, non_blocking=True), data['mask'].to(device, non_blocking=True)
        prompts, decoder_iters = _choose_prompts(config, data['prompt_point'], data['prompt_box'], data['prompt_mask'],
                                                 device)
        bad = ~torch.isfinite(images).all()
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
        boundary = micro % acc_steps == 0
        scaled = scaler.scale(loss) if scaler is not None else loss
        if boundary:
            scaled.backward()
        else:
            with model.no_sync():
                scaled.backward()

        packed = torch.cat([torch.stack([bad.float(), loss.detach().float()]), terms])
        all_reduce_sum_packed(packed, model, config.group)
        if carried_bad is not None:
            packed = torch.cat([torch.maximum(packed[0:1], carried_bad), packed[1:]])
        carried_bad = None if boundary else packed[0:1]

        if boundary:
            model.finish_gradient_sync()
            skip_flag = packed[0:1]
            if getattr(config, 'skip_inf_nan_grad', False) or scaler is not None:
                optimizer.check_finite()
                skip_flag = torch.maximum(skip_flag, optimizer.found_inf)
            inv_scale = scaler.state[2:3] if scaler is not None else None
            if clip_norm > 0:
                optimizer.clip_grad_norm_(clip_norm, inv_scale)
                inv_scale = None
            optimizer.step(inv_scale, skip_flag)
            if scaler is not None:
                scaler._found_inf = optimizer.found_inf
                scaler.update()
            optimizer.zero_grad()
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
