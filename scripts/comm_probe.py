"""World-of-one timing of the library communicator's collectives: what a 1-rank RCCL all-reduce costs by itself, so the
forced-sync bench (SAICV_DDP_FORCE_SYNC=1) can be read correctly."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleaicv_pytorch_training_examples_amd import engine  # noqa: E402

torch.cuda.set_device(0)
comm = engine.NativeComm(1, 0)
cur = torch.cuda.current_stream()
for mb in (4, 48):
    t = torch.randn(mb * (1 << 20) // 4, device='cuda')
    for _ in range(2):
        comm.allreduce_bucket(t, cur)
        comm.join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        comm.allreduce_bucket(t, cur)
        comm.join()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n
    print(f'all-reduce {mb} MiB, 1 rank: host {host * 1e6:.0f} us per call, end to end {tot * 1e6:.0f} us per call')
# does an in-flight 1-rank all-reduce slow an independent compute kernel?
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
t = torch.randn(48 * (1 << 20) // 4, device='cuda')


def gemms(with_comm):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        if with_comm and i % 5 == 0:
            comm.allreduce_bucket(t, cur)
        torch.matmul(a, b)
    if with_comm:
        comm.join()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


gemms(False)
print(f'20 GEMMs alone {gemms(False):.2f} ms; with four 48 MiB 1-rank all-reduces in flight + join {gemms(True):.2f} ms')
comm.close()
