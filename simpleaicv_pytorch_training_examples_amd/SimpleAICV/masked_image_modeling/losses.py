"""MAE reconstruction losses (reference SimpleAICV/masked_image_modeling/losses.py:11-45): mean over the REMOVED patches
of the per-patch mean squared / absolute error, fp32.  [N, L, p*p*3] tensors of a few MB: plain tensor arithmetic."""
import torch
import torch.nn as nn

__all__ = [
    'MSELoss',
    'L1Loss',
]


class MSELoss(nn.Module):

    def forward(self, pred, label, mask):
        pred, label, mask = pred.float(), label.float(), mask.float()
        per_patch = ((pred - label) ** 2).mean(dim=-1)
        return (per_patch * mask).sum() / (mask.sum() + 1e-4)


class L1Loss(nn.Module):

    def forward(self, pred, label, mask):
        pred, label, mask = pred.float(), label.float(), mask.float()
        return (torch.abs(pred - label) * mask).sum() / (mask.sum() + 1e-4)
