"""Per-parameter distance of the HIP fp32 gradients of resnet18cifar (batch 8) to the float64 CPU oracle, next to the fp32 oracle's:
where in the network the HIP backward loses accuracy (r04: 1e-3 relative L2 overall against 1e-6 for the CPU fp32 run)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones, losses  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'resnet18cifar'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
size = 32 if 'cifar' in name else 64
torch.manual_seed(0)
model = backbones.__dict__[name](num_classes=100)
sd = {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in model.state_dict().items()}
pnames = [n for n, _ in model.named_parameters()]
g = torch.Generator().manual_seed(1)
x = torch.randn(batch, 3, size, size, generator=g)
y = torch.randint(0, 100, (batch,), generator=g)
fwd = lambda leaves, inp: O.resnet_forward(name, leaves, inp, training=True)      # noqa: E731
_, _, g32 = O.loss_and_grads(fwd, sd, pnames, x, loss_fn=O.ce_loss, label=y)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
_, _, g64 = O.loss_and_grads(fwd, sd64, pnames, x.double(), loss_fn=O.ce_loss, label=y)
model = model.cuda()
crit = losses.CELoss()
logits = model(x.cuda())
crit(logits, y.cuda()).backward()
torch.cuda.synchronize()
print(f'{"parameter":44s} {"|g|":>10s} {"HIP vs f64":>11s} {"cpu32 vs f64":>12s}')
for n, p in model.named_parameters():
    r = g64[n]
    eh = float((p.grad.cpu().double() - r).norm() / r.norm().clamp_min(1e-300))
    ec = float((g32[n].double() - r).norm() / r.norm().clamp_min(1e-300))
    print(f'{n:44s} {float(r.norm()):10.3e} {eh:11.2e} {ec:12.2e}')
