"""Classification losses on the fused softmax-CE HIP kernel.

Mirrors reference SimpleAICV/classification/losses.py: CELoss (:14-28) and OneHotLabelCELoss
(:78-91).  Logits are taken in fp32 (`pred.float()` in the reference) and the loss is the
batch mean.  nn.CrossEntropyLoss's ignore_index is not used by the reference configs and is
not supported.
"""
import torch.nn as nn

from ... import ops

__all__ = [
    'CELoss',
    'OneHotLabelCELoss',
]


class CELoss(nn.Module):
    '''Cross Entropy Loss (hard int64 labels)'''

    def __init__(self):
        super(CELoss, self).__init__()

    def forward(self, pred, label):
        return ops.softmax_cross_entropy(pred, label, soft=False)


class OneHotLabelCELoss(nn.Module):
    '''Cross Entropy Loss, label is one-hot / soft (mixup, cutmix, smoothing)'''

    def __init__(self):
        super(OneHotLabelCELoss, self).__init__()

    def forward(self, pred, label):
        return ops.softmax_cross_entropy(pred, label, soft=True)
