"""SAM training losses on the MI355X HIP kernels -- drop-in for the reference module.

Interface contract (reference SimpleAICV/interactive_segmentation/losses.py): SAMLoss (:11) with the same
constructor arguments, `forward([all_iter_mask_preds, all_iter_iou_preds], targets)` -> dict with
'focal_loss', 'dice_loss', 'iou_predict_loss'; best-of-M mask selection on focal*w + dice*w (:99-115),
`supervise_all_iou`, per-iteration averaging and weights (:60-66).

Execution: everything that touches the [B, M, 1024, 1024] logits -- BCE, sigmoid, focal weighting, the dice
sums, the thresholded IoU counts -- is ONE streaming HIP pass producing six sums per (sample, mask)
(saicv_mask_loss_stats) and one pass for the gradient (saicv_mask_loss_grad); the reference issues ~15
full-resolution elementwise / reduction kernels per decoder iteration.  What remains is [B, M] arithmetic.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..._lib import check, dtype_code, lib, ptr, require_gpu, stream

__all__ = [
    'SAMLoss',
]


class MaskLossStatsFn(torch.autograd.Function):
    """logits [B, M, H, W] (bf16 / fp32), targets [B, 1, H, W] -> stats [B, M, 6] fp32:
    sum focal, sum sigmoid*t, sum sigmoid, sum t, #(x>thr & t>thr), #(x>thr | t>thr)."""

    @staticmethod
    def forward(ctx, logits, targets, alpha, gamma, thr):
        require_gpu(logits, targets)
        b, m, h, w = logits.shape
        if targets.shape[0] != b or targets.numel() != b * h * w:
            raise ValueError(f'targets {tuple(targets.shape)} do not broadcast over logits {tuple(logits.shape)}')
        logits = logits.contiguous()
        targets = targets.contiguous().float()
        stats = torch.empty((b, m, 6), dtype=torch.float32, device=logits.device)
        check(lib().saicv_mask_loss_stats(dtype_code(logits.dtype), ptr(logits), ptr(targets), ptr(stats), b, m, h * w,
                                          float(alpha), float(gamma), float(thr), stream()), 'mask_loss_stats')
        ctx.save_for_backward(logits, targets)
        ctx.cfg = (float(alpha), float(gamma))
        return stats

    @staticmethod
    def backward(ctx, dstats):
        logits, targets = ctx.saved_tensors
        alpha, gamma = ctx.cfg
        b, m, h, w = logits.shape
        coef = dstats[..., :3].float().contiguous()
        dlogits = torch.empty_like(logits)
        check(lib().saicv_mask_loss_grad(dtype_code(logits.dtype), ptr(logits), ptr(targets), ptr(coef), ptr(dlogits), b,
                                         m, h * w, alpha, gamma, stream()), 'mask_loss_grad')
        return dlogits, None, None, None, None


class UpMaskLossStatsFn(torch.autograd.Function):
    """MaskLossStatsFn over the x4 bilinear upsampling of LOW-resolution logits [B, M, h, w] (targets [B, 1, 4h, 4w]):
    the full-resolution logits are interpolated in registers, forward and backward; the gradient comes out with
    respect to the low-resolution logits (reference sam.py:155-158 + losses.py:136-198 in two kernels)."""

    @staticmethod
    def forward(ctx, low, targets, alpha, gamma, thr):
        require_gpu(low, targets)
        b, m, h, w = low.shape
        if targets.shape[0] != b or targets.numel() != b * 16 * h * w:
            raise ValueError(f'targets {tuple(targets.shape)} are not the x4 grid of logits {tuple(low.shape)}')
        low = low.contiguous()
        targets = targets.contiguous().float()
        stats = torch.empty((b, m, 6), dtype=torch.float32, device=low.device)
        check(lib().saicv_mask_loss_stats_up4(dtype_code(low.dtype), ptr(low), ptr(targets), ptr(stats), b, m, h, w,
                                              float(alpha), float(gamma), float(thr), stream()), 'mask_loss_stats_up4')
        ctx.save_for_backward(low, targets)
        ctx.cfg = (float(alpha), float(gamma))
        return stats

    @staticmethod
    def backward(ctx, dstats):
        low, targets = ctx.saved_tensors
        alpha, gamma = ctx.cfg
        b, m, h, w = low.shape
        coef = dstats[..., :3].float().contiguous()
        dlow = torch.empty_like(low)
        check(lib().saicv_mask_loss_grad_up4(dtype_code(low.dtype), ptr(low), ptr(targets), ptr(coef), ptr(dlow), b, m, h, w,
                                             alpha, gamma, stream()), 'mask_loss_grad_up4')
        return dlow, None, None, None, None


class SAMLoss(nn.Module):

    def __init__(self, alpha=0.25, gamma=2, focal_loss_weight=20, dice_loss_weight=1, iou_predict_loss_weight=1,
                 supervise_all_iou=True, mask_threshold=0.):
        super(SAMLoss, self).__init__()
        self.alpha = alpha
        self.gamma = gamma
        self.focal_loss_weight = focal_loss_weight
        self.dice_loss_weight = dice_loss_weight
        self.iou_predict_loss_weight = iou_predict_loss_weight
        self.supervise_all_iou = supervise_all_iou
        self.mask_threshold = mask_threshold

    def forward(self, all_iter_preds, targets):
        all_iter_mask_preds, all_iter_iou_preds = all_iter_preds
        assert len(all_iter_mask_preds) == len(all_iter_iou_preds)
        focal_loss, dice_loss, iou_predict_loss = 0., 0., 0.
        iter_num = len(all_iter_mask_preds)
        for per_iter_mask_preds, per_iter_iou_preds in zip(all_iter_mask_preds, all_iter_iou_preds):
            f, d, i = self.compute_per_iter_loss(per_iter_mask_preds, per_iter_iou_preds, targets)
            focal_loss = focal_loss + f
            dice_loss = dice_loss + d
            iou_predict_loss = iou_predict_loss + i
        return {
            'focal_loss': focal_loss / float(iter_num) * self.focal_loss_weight,
            'dice_loss': dice_loss / float(iter_num) * self.dice_loss_weight,
            'iou_predict_loss': iou_predict_loss / float(iter_num) * self.iou_predict_loss_weight,
        }

    def per_mask_losses(self, mask_preds, iou_preds, targets):
        """-> focal [B, M], dice [B, M], iou-prediction [B, M], each already divided by the batch size
        (reference focal_loss :136-153, dice_loss :155-176, iou_predict_loss :178-198)."""
        batch_size = mask_preds.shape[0]
        hw = mask_preds.shape[2] * mask_preds.shape[3]
        low = getattr(mask_preds, '_saicv_low', None)
        if low is not None and low.shape[:2] == mask_preds.shape[:2] and low.shape[2] * 4 == mask_preds.shape[2]:
            # the model handed its low-resolution logits along (sam.py): loss and gradient without the full-resolution tensor
            stats = UpMaskLossStatsFn.apply(low, targets, self.alpha, self.gamma, self.mask_threshold)
        else:
            stats = MaskLossStatsFn.apply(mask_preds, targets, self.alpha, self.gamma, self.mask_threshold)
        focal = stats[..., 0] / float(hw) / batch_size
        dice = (1. - (2. * stats[..., 1] + 1) / (stats[..., 2] + stats[..., 3] + 1)) / batch_size
        with torch.no_grad():
            gt_ious = torch.clamp(stats[..., 4] / torch.clamp(stats[..., 5], min=1e-6), min=0.0, max=1.0)
        iou = F.mse_loss(iou_preds.float(), gt_ious, reduction="none") / batch_size
        return focal, dice, iou

    def compute_per_iter_loss(self, per_iter_mask_preds, per_iter_iou_preds, targets):
        focal_loss, dice_loss, iou_predict_loss = self.per_mask_losses(per_iter_mask_preds, per_iter_iou_preds, targets)
        # several masks per object: focal / dice only for the best one, IoU head for all of them (or the best)
        if focal_loss.shape[1] > 1:
            combine_loss = focal_loss * self.focal_loss_weight + dice_loss * self.dice_loss_weight
            best_index = torch.argmin(combine_loss, dim=-1)
            batch_index = torch.arange(combine_loss.shape[0], device=combine_loss.device)
            focal_loss = focal_loss[batch_index, best_index].unsqueeze(1)
            dice_loss = dice_loss[batch_index, best_index].unsqueeze(1)
            if self.supervise_all_iou:
                iou_predict_loss = iou_predict_loss.mean(dim=-1, keepdim=True)
            else:
                iou_predict_loss = iou_predict_loss[batch_index, best_index].unsqueeze(1)
        return focal_loss.sum(), dice_loss.sum(), iou_predict_loss.sum()
