"""DETR prediction head on the MI355X HIP kernels (reference SimpleAICV/detection/models/head.py:184-214):
class logits Linear(256 -> num_classes) and a 3-layer box MLP whose sigmoid runs in fp32; same parameter
names (`cls_head.*`, `reg_head.{0,2,4}.*`) and xavier init order."""
import torch
import torch.nn as nn

from .... import ops_tfm


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
