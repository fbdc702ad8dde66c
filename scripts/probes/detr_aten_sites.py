"""r06: which Python lines of the DETR step launch its ATen glue kernels?  Runs bench.loop_workload('resnet50_detr_config') eagerly under
torch.profiler (with_stack) for two steps after warm-up and prints, per (ATen op, innermost frame inside this package), the number
of device kernels and their time in one step -- the input of a launch diet."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.set_device(0)
    args = argparse.Namespace(batch=8)
    # the STATIC form of the step (device-side assignment, padded pair buffers: what the captured step replays), launched eagerly:
    # StepGraph stays eager when it believes graph packet capture is on
    import warnings
    import simpleaicv_pytorch_training_examples_amd as pkg
    pkg.GRAPH_PACKET_CAPTURE_OFF = False
    warnings.simplefilter('ignore')
    run, *_ = bench.loop_workload('resnet50_detr_config', args, 1, 0, torch.device('cuda', 0), use_graph=True)
    run(3)
    torch.cuda.synchronize()
    steps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        run(steps)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.kernels or not (ev.name.startswith('aten::') or ev.name.startswith('autograd::')):
            continue
        # leaf ops only: an op whose kernels are also counted by a child op (aten::to -> aten::copy_) would count twice
        if any(c.kernels for c in ev.cpu_children):
            continue
        site = next((f for f in ev.stack if 'simpleaicv_pytorch_training_examples_amd' in f and 'site-packages' not in f), None)
        if site is None:
            site = 'autograd / other: ' + (ev.stack[0] if ev.stack else '?')
        site = site.replace(ROOT + '/', '').replace('simpleaicv_pytorch_training_examples_amd/', '')
        a = agg[(ev.name, site)]
        a[0] += len(ev.kernels)
        a[1] += sum(k.duration for k in ev.kernels)
    tot_n = sum(a[0] for a in agg.values()) / steps
    tot_t = sum(a[1] for a in agg.values()) / steps
    print(f'ATen device kernels per step: {tot_n:.0f}, {tot_t / 1e3:.2f} ms')
    for (op, site), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f'{a[0] / steps:6.1f} {a[1] / steps:8.1f} us  {op:34s} {site[:150]}')


if __name__ == '__main__':
    main()
