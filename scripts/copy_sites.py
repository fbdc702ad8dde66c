"""Which Python lines issue aten::copy_ / aten::_to_copy during an eager ResNet-50 step (torch.profiler, stacks)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    sys.argv = ['bench.py', '--model', 'resnet50', '--steps', '2', '--warmup', '2', '--no-cpu-baseline', '--no-secondary', '--max-windows', '1',
                '--no-kernel-timer', '--eager']
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        try:
            bench.main()
        except SystemExit:
            pass
    sites = collections.Counter()
    for e in prof.events():
        if e.name in ('aten::copy_', 'aten::_to_copy', 'aten::clone', 'aten::contiguous'):
            st = [s for s in (e.stack or []) if 'simpleaicv_pytorch_training_examples_amd' in s or 'bench.py' in s]
            key = (e.name, str(e.input_shapes)[:60], st[0].split('simpleaicv_pytorch_training_examples_amd/')[-1][:90] if st else 'no package frame')
            sites[key] += 1
    for (name, shp, where), c in sites.most_common(30):
        print(f'{c:5d} {name:16s} {shp:60s} {where}')


if __name__ == '__main__':
    main()
