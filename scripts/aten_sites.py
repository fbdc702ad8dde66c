"""Which lines of the package issue the small ATen ops of a training step (forward side): a TorchDispatchMode counts every
aten op and charges it to the innermost frame inside the package.  python scripts/aten_sites.py sam_b 4"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SITES = collections.Counter()
OPS = collections.Counter()
ACTIVE = [False]


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if ACTIVE[0]:
            name = str(func)
            if not any(s in name for s in ('view', 'detach', 'alias', 'as_strided', 'select', 'slice', 'expand', 'permute', 't.default',
                                           'transpose', 'unsqueeze', 'squeeze', 'reshape', '_unsafe_view', 'unbind', 'split')):
                site = None
                for fr in reversed(traceback.extract_stack(limit=14)[:-1]):
                    if 'simpleaicv_pytorch_training_examples_amd' in fr.filename and 'aten_sites' not in fr.filename:
                        site = f'{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}'
                        break
                SITES[site or 'other'] += 1
                OPS[name] += 1
        return func(*args, **(kwargs or {}))


def main():
    model, batch = sys.argv[1], sys.argv[2]
    sys.argv = ['bench.py', '--model', model, '--batch', batch, '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-secondary',
                '--max-windows', '1', '--no-kernel-timer', '--eager']
    with Count():
        ACTIVE[0] = True
        try:
            bench.main()
        except SystemExit:
            pass
        ACTIVE[0] = False
    n = sum(SITES.values())
    print(f'{n} kernel-launching aten ops seen from the main thread (3 steps + set-up)')
    for s, c in SITES.most_common(45):
        print(f'{c:6d}  {s}')
    print('--- ops')
    for s, c in OPS.most_common(25):
        print(f'{c:6d}  {s}')


if __name__ == '__main__':
    main()
