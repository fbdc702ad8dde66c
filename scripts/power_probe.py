"""Is the chip at its power limit while the GEMM kernels run?  Runs one launch shape in a loop for a few seconds per case while a thread
samples `rocm-smi --showpower --showclocks --json` and prints average socket power, shader clock and throughput per case.
    python scripts/power_probe.py [lib tag ...]        (library variants from scripts/build_variant_lib.py; default: product library)"""
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            c = d.get('card0', {})
            out.append({k: v for k, v in c.items() if 'ower' in k or 'sclk' in k or 'mclk' in k or 'fclk' in k})
        except Exception as e:      # noqa: BLE001
            out.append({'error': str(e)[:80]})
        time.sleep(0.15)


def main():
    import torch
    from simpleaicv_pytorch_training_examples_amd import _lib
    tag = sys.argv[1] if len(sys.argv) > 1 else '-'
    if tag != '-':
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f'libsaicv_hip_{tag}.so')
    from simpleaicv_pytorch_training_examples_amd import ops
    from simpleaicv_pytorch_training_examples_amd._lib import check, lib, ptr
    L, st = lib(), _lib.stream()
    bf = torch.bfloat16
    M = 50432

    def lin(K, N, zero=False):
        x = (torch.zeros if zero else torch.randn)(M, K, device='cuda').to(bf)
        w = ((torch.zeros if zero else torch.randn)(N, K, device='cuda') * 0.03).to(bf)
        b = torch.randn(N, device='cuda')
        y = torch.empty(M, N, device='cuda', dtype=bf)
        return (lambda: check(L.saicv_linear_fwd(0, ptr(x), ptr(w), ptr(b), ptr(y), M, K, N, 0, 0, 0, 1, st))), 2.0 * M * K * N, (x, w, b, y)

    def copy():
        a = torch.randn(1 << 28, device='cuda').to(bf)
        b_ = torch.empty_like(a)
        return (lambda: b_.copy_(a)), 0.0, (a, b_)

    cases = [('idle', None), ('linear fc1 K=768 N=3072 random', lambda: lin(768, 3072)), ('linear fc1 zeros', lambda: lin(768, 3072, True)),
             ('linear fc2 K=3072 N=768 random', lambda: lin(3072, 768)), ('linear proj K=768 N=768 random', lambda: lin(768, 768)),
             ('copy 512 MiB bf16', copy)]
    for name, mk in cases:
        if mk is None:
            stop, out = threading.Event(), []
            th = threading.Thread(target=sample, args=(stop, out))
            th.start()
            time.sleep(1.5)
            stop.set()
            th.join()
            print(json.dumps({'case': name, 'lib': tag, 'samples': out[-3:]}), flush=True)
            continue
        fn, flops, keep = mk()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < 3.0:
            for _ in range(50):
                fn()
            n += 50
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(json.dumps({'case': name, 'lib': tag, 'us_per_launch': round(us, 1), 'tflops': round(flops / us / 1e6, 1), 'samples': out[2:][-4:]}), flush=True)
        del keep


if __name__ == '__main__':
    main()
