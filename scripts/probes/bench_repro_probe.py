"""r06: does the bench's ResNet-50 loop repeat bit for bit ACROSS processes in deterministic mode?  Each child process builds
bench.classification_workload and prints, per step, the loss, a bit hash of the input batch and of all weights."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(graph):
    import torch
    import bench
    from simpleaicv_pytorch_training_examples_amd import ops
    torch.cuda.set_device(0)
    args = argparse.Namespace(batch=256)
    run, model, scaler, state, info, batch, size = bench.classification_workload('resnet50', args, 1, 0, torch.device('cuda', 0), graph)
    seen = []
    h = model.register_forward_pre_hook(lambda m, i: seen.append(int(i[0].detach().float().view(torch.int32).to(torch.int64).sum())) if not torch.cuda.is_current_stream_capturing() else None)
    import simpleaicv_pytorch_training_examples_amd as pkg
    print(json.dumps({'deterministic': bool(ops.is_deterministic()), 'packet_capture_off': bool(pkg.GRAPH_PACKET_CAPTURE_OFF)}), flush=True)
    for s in range(10):
        run(1)
        torch.cuda.synchronize()
        w = sum(int(p.detach().view(torch.int32).to(torch.int64).sum()) for p in model.parameters())
        print(json.dumps({'step': s, 'loss': state['loss'], 'weights': w, 'input': seen[-1] if seen else None, 'scale': scaler.get_scale() if scaler else None}), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'child':
        child(sys.argv[2] == 'graph')
        sys.exit(0)
    # variants: "<name>@<ENV=V,ENV=V>" ... (a leg with packet capture ON also needs SAICV_STEP_GRAPH_WITH_PACKETS=1, or StepGraph stays eager); every variant runs the graph form REPS times and is compared with one eager run
    variants = sys.argv[1:] or ['default@']
    reps = int(os.environ.get('REPS', '2'))

    def one(mode, env):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', mode], capture_output=True, text=True, env=env)
        if p.returncode:
            print(mode, 'rc', p.returncode, p.stderr[-1500:])
        return [l for l in p.stdout.splitlines() if l.startswith('{')]

    for v in variants:
        name, _, envs = v.partition('@')
        env = dict(os.environ, **dict(kv.split('=', 1) for kv in envs.split(',') if kv))
        ref = one('eager', env)
        eager2 = one('eager', env)
        print(f'[{name}] eager: two processes identical: {ref == eager2}')
        for r in range(reps):
            g = one('graph', env)
            first = next((i for i, (a, b) in enumerate(zip(ref, g)) if a != b), None)
            print(f'[{name}] graph run {r}: identical to eager: {g == ref}' + ('' if first is None else f'  first difference at line {first}: {g[first][:120]} (eager {ref[first][:120]})'))
        if name == variants[0].partition('@')[0]:
            for a in ref:
                print('  ', a)
