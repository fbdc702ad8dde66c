"""GPU timeline (torch.profiler / roctracer) of eager ResNet-50 steps with the data-parallel machinery forced on in a world of
one: where the compute queue idles.  SAICV_DDP_FORCE_SYNC=1 python scripts/ddp_timeline.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    sys.argv = ['bench.py', '--model', 'resnet50', '--steps', '4', '--warmup', '3', '--no-cpu-baseline', '--no-secondary', '--max-windows', '1',
                '--no-kernel-timer', '--eager']
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        try:
            bench.main()
        except SystemExit:
            pass
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ev.sort(key=lambda e: e.time_range.start)
    print('kernels', len(ev))
    ev = ev[-1500:]
    t0 = ev[0].time_range.start
    busy = 0.0
    prev_end = None
    gaps = []
    streams = {}
    for e in ev:
        s, en = e.time_range.start - t0, e.time_range.end - t0
        busy += en - s
        streams[getattr(e, 'stream', None)] = streams.get(getattr(e, 'stream', None), 0) + 1
        if prev_end is not None and s - prev_end > 60:
            gaps.append((s - prev_end, s, e.name[:60], prev_name))
        if prev_end is None or en > prev_end:
            prev_end, prev_name = en, e.name[:50]
    import collections, re
    cat = collections.Counter()
    cnt = collections.Counter()
    for e in ev:
        m = re.search(r'(igemm_nt|igemm_tn|bn_bwd_apply|bn_act_fwd|bn_bwd_reduce|bn_finalize|bn_reduce_partials|oneRankReduce|maxpool|pack_weight|pack_input|sgd|Memcpy|Memset|at::native)', e.name)
        k = m.group(1) if m else e.name[:30]
        cat[k] += e.time_range.end - e.time_range.start
        cnt[k] += 1
    for k, v in cat.most_common(14):
        print(f'   {k:24s} {v / 1e3:9.2f} ms over {cnt[k]:5d} launches  ({v / cnt[k]:8.1f} us each)')
    span = ev[-1].time_range.end - t0
    print(f'span {span / 1e3:.2f} ms, sum of kernel durations {busy / 1e3:.2f} ms, gaps > 60 us: {len(gaps)} totalling {sum(g[0] for g in gaps) / 1e3:.2f} ms')
    for g in sorted(gaps, reverse=True)[:25]:
        print(f'  gap {g[0]:8.1f} us at {g[1] / 1e3:8.2f} ms before {g[2]}  (after {g[3]})')


if __name__ == '__main__':
    main()
