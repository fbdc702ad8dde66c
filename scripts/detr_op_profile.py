"""ATen op counts of one DETR (reference config) training step: which host-side ops make the step host-bound."""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = ['bench.py', '--model', sys.argv[1] if len(sys.argv) > 1 else 'resnet50_detr_config', '--batch', sys.argv[2] if len(sys.argv) > 2 else '8',
                '--steps', '2', '--warmup', '2', '--eager', '--no-cpu-baseline', '--no-secondary', '--max-windows', '1', '--no-kernel-timer']
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
        try:
            bench.main()
        except SystemExit:
            pass
    ev = prof.key_averages(group_by_stack_n=4)
    rows = []
    for e in ev:
        if e.key.startswith('aten::') and e.count >= 8:
            rows.append((e.count, e.self_cpu_time_total / 1e3, e.key, ' <- '.join(s.split('/')[-1] for s in e.stack[:3])))
    rows.sort(reverse=True)
    for c, ms, k, st in rows[:70]:
        print(f'{c:6d} {ms:8.2f} ms  {k:32s} {st[:150]}')


if __name__ == '__main__':
    main()
