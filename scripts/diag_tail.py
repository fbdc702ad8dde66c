"""Bisect the fp32 gradient discrepancy of ResNet18Cifar b64 from the top: (1) avgpool + fc + CE tail, (2) a ConvBnAct
block at layer4's shape, each against fp64 on the CPU."""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import rel_err
from simpleaicv_pytorch_training_examples_amd import ops
from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import losses

g = torch.Generator().manual_seed(0)
# ---- (1) tail
z = torch.relu(torch.randn(64, 512, 4, 4, generator=g))
w = torch.randn(100, 512, generator=g) * 0.05
b = torch.zeros(100)
y = torch.randint(0, 100, (64,), generator=g)
zr = z.double().requires_grad_(True); wr = w.double().requires_grad_(True)
lr = F.cross_entropy(F.linear(zr.mean((2, 3)), wr, b.double()), y); lr.backward()
zd = z.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
wd = w.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True)
pooled = ops.global_avg_pool(zd)
pooled.retain_grad()
logits = ops.linear(pooled, wd, bd, out_f32=True)
logits.retain_grad()
ld = losses.CELoss()(logits, y.cuda()); ld.backward(); torch.cuda.synchronize()
print('tail: loss', float(ld), float(lr))
zr2 = z.double().requires_grad_(True)
pr = zr2.mean((2, 3)); pr.retain_grad(); lg = F.linear(pr, w.double(), b.double()); lg.retain_grad()
F.cross_entropy(lg, y).backward()
print('  dlogits', rel_err(logits.grad, lg.grad), ' dpooled', rel_err(pooled.grad, pr.grad), ' dz', rel_err(zd.grad, zr.grad), ' dW', rel_err(wd.grad, wr.grad))

# ---- (2) one conv-bn-relu block with a residual at layer4 shape, fp32
for (n, c, h, k, r, s) in [(64, 512, 4, 512, 3, 1), (64, 256, 8, 512, 3, 2), (64, 256, 8, 512, 1, 2), (64, 64, 32, 64, 3, 1)]:
    pad = r // 2
    x = torch.randn(n, c, h, h, generator=g)
    wt = torch.randn(k, c, r, r, generator=g) * (2.0 / (c * r * r)) ** 0.5
    oh = (h + 2 * pad - r) // s + 1
    res = torch.randn(n, k, oh, oh, generator=g)
    dz = torch.randn(n, k, oh, oh, generator=g)
    xr = x.double().requires_grad_(True); wr = wt.double().requires_grad_(True); rr = res.double().requires_grad_(True)
    gr = torch.ones(k, dtype=torch.double, requires_grad=True); br = torch.zeros(k, dtype=torch.double, requires_grad=True)
    yr = F.conv2d(xr, wr, None, s, pad)
    zz = F.relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5) + rr)
    zz.backward(dz.double())
    bn = torch.nn.BatchNorm2d(k).cuda()
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = wt.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rd = res.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    zd = ops.conv_bn_act(xd, wd, bn, s, pad, True, rd)
    zd.backward(dz.cuda().contiguous(memory_format=torch.channels_last)); torch.cuda.synchronize()
    print(f'block {(n, c, h, k, r, s)}: z {rel_err(zd, zz):.1e} dx {rel_err(xd.grad, xr.grad):.1e} dw {rel_err(wd.grad, wr.grad):.1e} '
          f'dgamma {rel_err(bn.weight.grad, gr.grad):.1e} dbeta {rel_err(bn.bias.grad, br.grad):.1e} dres {rel_err(rd.grad, rr.grad):.1e}')
