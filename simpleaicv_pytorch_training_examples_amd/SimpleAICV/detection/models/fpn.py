"""RetinaNet's feature pyramid on the MI355X kernels (reference SimpleAICV/detection/models/fpn.py:14-82).

Same module tree (`P3_1 ... P5_2`, `P6`, `P7 = Sequential(ReLU, Conv2d)`), constructor arguments and construction order as the
reference, so a seeded construction draws identical initial weights and checkpoints load key for key.  Every convolution
(lateral 1x1, smoothing 3x3, the two stride-2 3x3 of P6 / P7, all with bias) is the implicit-GEMM kernel through
`ops.conv2d`; the top-down merge (bilinear resize to the finer level + add) and the lone ReLU are two small elementwise
ops per level on NHWC tensors."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops


def _conv(x, m):
    return ops.conv2d(x, m.weight, m.bias, m.stride[0], m.padding[0])


class RetinaFPN(nn.Module):

    def __init__(self, inplanes, planes, use_p5=False):
        super(RetinaFPN, self).__init__()
        self.use_p5 = use_p5                     # inplanes: [C3, C4, C5] channel counts
        for level, cin in zip((3, 4, 5), inplanes):
            setattr(self, f'P{level}_1', nn.Conv2d(cin, planes, kernel_size=1, stride=1, padding=0))
            setattr(self, f'P{level}_2', nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1))
        self.P6 = nn.Conv2d(planes if use_p5 else inplanes[2], planes, kernel_size=3, stride=2, padding=1)
        self.P7 = nn.Sequential(nn.ReLU(), nn.Conv2d(planes, planes, kernel_size=3, stride=2, padding=1))

    @staticmethod
    def _merge(top, lateral):
        if top.is_cuda and top.shape[1] % 4 == 0:
            return ops.resize_bilinear_add(top, lateral)      # one pass; its backward is a fixed-order gather (ATen's uses atomics)
        return F.interpolate(top, size=(lateral.shape[2], lateral.shape[3]), mode='bilinear') + lateral

    def forward(self, inputs):
        C3, C4, C5 = inputs
        P5 = _conv(C5, self.P5_1)
        P4 = self._merge(P5, _conv(C4, self.P4_1))
        P3 = self._merge(P4, _conv(C3, self.P3_1))
        P5, P4, P3 = _conv(P5, self.P5_2), _conv(P4, self.P4_2), _conv(P3, self.P3_2)
        P6 = _conv(P5 if self.use_p5 else C5, self.P6)
        P7 = _conv(torch.relu(P6), self.P7[1])
        return [P3, P4, P5, P6, P7]
