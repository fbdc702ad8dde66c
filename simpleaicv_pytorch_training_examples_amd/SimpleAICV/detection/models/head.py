"""Detection heads on the MI355X HIP kernels (reference SimpleAICV/detection/models/head.py).
DETR (:184-214): class logits Linear(256 -> num_classes) and a 3-layer box MLP whose sigmoid runs in fp32; same parameter
names (`cls_head.*`, `reg_head.{0,2,4}.*`) and xavier init order.
RetinaNet (:15-86) and FCOS (:88-181), SURVEY.md 8(f) rank 2: towers of 3x3 convolutions (+ GroupNorm for FCOS) + ReLU shared
by all pyramid levels and 3x3 output convolutions; same module trees (`cls_head.{0,2,4,6}`, `cls_out`, ...), N(0, 0.01)
weights, zero biases and the focal-loss prior on `cls_out.bias`, drawn in the reference's order.  The convolutions run on the
implicit-GEMM kernel (`ops.conv2d`), GroupNorm + ReLU on `csrc/groupnorm.hip` (`ops.group_norm`); the lone ReLUs and the fp32
sigmoids are elementwise ops on the NHWC tensors."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops, ops_tfm


def _tower(x, seq):
    """Conv2d (-> GroupNorm) -> ReLU stacks of a head on NHWC data.  GroupNorm and the ReLU behind it are one node
    (ops.group_norm, csrc/groupnorm.hip); channel counts its 16-byte chunks do not divide fall back to the tensor op."""
    layers = list(seq)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, nn.Conv2d):
            x = ops.conv2d(x, layer.weight, layer.bias, layer.stride[0], layer.padding[0])
        elif isinstance(layer, nn.GroupNorm):
            fuse = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
            if layer.num_channels % 8 == 0:
                x = ops.group_norm(x, layer, relu=fuse)
                i += int(fuse)
            else:
                x = F.group_norm(x, layer.num_groups, layer.weight, layer.bias, layer.eps)
        else:
            x = torch.relu(x)
        i += 1
    return x


def _init_convs(module, cls_out=None, prior=0.01):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, std=0.01)
            if m.bias is not None:
                nn.init.constant_(m.bias, val=0)
    if cls_out is not None:
        cls_out.bias.data.fill_(-math.log((1 - prior) / prior))


def _conv3(cin, cout, bias=True):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=bias)


class RetinaClsHead(nn.Module):

    def __init__(self, inplanes, num_anchors, num_classes, num_layers=4):
        super(RetinaClsHead, self).__init__()
        layers = []
        for _ in range(num_layers):
            layers += [_conv3(inplanes, inplanes), nn.ReLU(inplace=True)]
        self.cls_head = nn.Sequential(*layers)
        self.cls_out = _conv3(inplanes, num_anchors * num_classes)
        self.sigmoid = nn.Sigmoid()
        _init_convs(self, self.cls_out)

    def forward(self, x):
        x = _tower(_tower(x, self.cls_head), [self.cls_out])
        return self.sigmoid(x.float())


class RetinaRegHead(nn.Module):

    def __init__(self, inplanes, num_anchors, num_layers=4):
        super(RetinaRegHead, self).__init__()
        layers = []
        for _ in range(num_layers):
            layers += [_conv3(inplanes, inplanes), nn.ReLU(inplace=True)]
        self.reg_head = nn.Sequential(*layers)
        self.reg_out = _conv3(inplanes, num_anchors * 4)
        _init_convs(self)

    def forward(self, x):
        return _tower(_tower(x, self.reg_head), [self.reg_out])


class FCOSClsRegCntHead(nn.Module):

    def __init__(self, inplanes, num_classes, num_layers=4, use_gn=True, cnt_on_reg=True):
        super(FCOSClsRegCntHead, self).__init__()
        self.cnt_on_reg = cnt_on_reg

        def tower():
            layers = []
            for _ in range(num_layers):
                layers.append(_conv3(inplanes, inplanes, bias=use_gn is False))
                if use_gn:
                    layers.append(nn.GroupNorm(32, inplanes))
                layers.append(nn.ReLU(inplace=True))
            return nn.Sequential(*layers)

        self.cls_head = tower()
        self.reg_head = tower()
        self.cls_out = _conv3(inplanes, num_classes)
        self.reg_out = _conv3(inplanes, 4)
        self.center_out = _conv3(inplanes, 1)
        self.sigmoid = nn.Sigmoid()
        _init_convs(self, self.cls_out)

    def forward(self, x):
        cls_x, reg_x = _tower(x, self.cls_head), _tower(x, self.reg_head)
        cls_output = _tower(cls_x, [self.cls_out])
        reg_output = _tower(reg_x, [self.reg_out])
        center_output = _tower(reg_x if self.cnt_on_reg else cls_x, [self.center_out])
        return self.sigmoid(cls_output.float()), reg_output, self.sigmoid(center_output.float())


class DETRClsRegHead(nn.Module):

    def __init__(self, hidden_inplanes, num_classes, num_layers=3):
        super(DETRClsRegHead, self).__init__()
        self.cls_head = nn.Linear(hidden_inplanes, num_classes)
        reg_layers = []
        for _ in range(num_layers - 1):
            reg_layers.append(nn.Linear(hidden_inplanes, hidden_inplanes))
            reg_layers.append(nn.ReLU(inplace=True))
        reg_layers.append(nn.Linear(hidden_inplanes, 4))
        self.reg_head = nn.Sequential(*reg_layers)
        self.sigmoid = nn.Sigmoid()
        for m in self.parameters():
            if m.dim() > 1:
                nn.init.xavier_uniform_(m)

    def forward(self, x):
        cls_output = ops_tfm.linear_nd(x, self.cls_head.weight, self.cls_head.bias)
        r = x
        for layer in self.reg_head:
            r = ops_tfm.linear_nd(r, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else torch.relu(r)
        reg_output = self.sigmoid(r.float())
        return cls_output, reg_output
