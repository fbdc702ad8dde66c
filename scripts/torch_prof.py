"""Maps device kernels / memcpys back to the Python-side op that issued them (torch.profiler),
to hunt stray copies and tiny kernels in a training step."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

model, crit, soft, engine = bench.build('resnet50', torch.device('cuda', 0))
opt = bench.make_optimizer('resnet50', model, engine)
ddp = engine.DistributedDataParallel(model)
scaler = engine.GradScaler(device='cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
images = torch.randn(B, 224, 224, 3, device='cuda').permute(0, 3, 1, 2)
labels = torch.randint(0, 1000, (B,), device='cuda')


def step():
    opt.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = crit(ddp(images), labels)
    scaler.scale(loss).backward()
    ddp.finish_gradient_sync()
    scaler.step(opt)
    scaler.update()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='count', row_limit=45, max_name_column_width=60))
