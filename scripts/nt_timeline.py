"""Per-workgroup phase timeline of igemm_nt1_kernel (debug build of the library with -DSAICV_NT_TIMELINE, built by
`python scripts/nt_timeline.py --build` HERE before the GPU call; the product library is untouched).

For a handful of representative launches (ViT-B linears, ResNet-50 convolutions) every workgroup records shader-clock stamps
  0 entry | 1 prologue DMA issued | 2 first K step landed (first barrier passed) | 3 K loop done | 4 tile staged in LDS |
  5 statistics done | 6 stores issued | 7 stores acknowledged
plus its XCC / HW_ID (CU) and the 100 MHz real-time counter at exit.  Output: one JSON line per launch with the mean duration of each
phase in microseconds, the mean workgroup lifetime, the mean gap between consecutive workgroups of one CU slot, and the launch wall time."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, 'simpleaicv_pytorch_training_examples_amd')
TL_LIB = os.path.join(PKG, 'libsaicv_hip_tl.so')


def build():
    from simpleaicv_pytorch_training_examples_amd import build as b
    b.build()
    objdir = os.path.join(b.CSRC, 'build')
    o = os.path.join(objdir, 'igemm_tl.o')
    subprocess.check_call([b._hipcc()] + b.FLAGS + ['-DSAICV_NT_TIMELINE', '-c', os.path.join(b.CSRC, 'igemm.hip'), '-o', o])
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in b.SOURCES if s != 'igemm.hip'] + [o]
    subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', TL_LIB] + objs + ['-ldl'])
    print(TL_LIB)


def main():
    import torch
    from simpleaicv_pytorch_training_examples_amd import _lib
    _lib.LIB_PATH = TL_LIB
    from simpleaicv_pytorch_training_examples_amd import ops
    from simpleaicv_pytorch_training_examples_amd._lib import check, lib, ptr
    L, st = lib(), _lib.stream()
    raw = ctypes.CDLL(TL_LIB)
    raw.saicv_debug_nt_timeline.argtypes = [ctypes.c_void_p]
    raw.saicv_debug_nt_timeline.restype = None
    bf = torch.bfloat16
    cap = 1 << 17
    buf = torch.zeros(cap * 16, dtype=torch.int64, device='cuda')

    def measure(name, fn, flops):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        buf.zero_()
        raw.saicv_debug_nt_timeline(ctypes.c_void_p(buf.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        raw.saicv_debug_nt_timeline(ctypes.c_void_p(0))
        wall = e0.elapsed_time(e1) * 1e3
        r = buf.view(cap, 16).cpu()
        r = r[r[:, 0] != 0]
        n = r.shape[0]
        if n == 0:
            print(json.dumps({'launch': name, 'error': 'no records'}))
            return
        t = r[:, :8].double()
        # the shader clock is per XCD (unsynchronised counters): only differences inside one workgroup are used; the 100 MHz real-time
        # counter orders workgroups across the chip
        rt0, rt1 = r[:, 10].double() / 100.0, r[:, 9].double() / 100.0          # us
        mhz = float((t[:, 7] - t[:, 0]).sum() / (rt1 - rt0).sum())
        d = (t[:, 1:] - t[:, :-1]) / mhz
        life = rt1 - rt0
        hw = r[:, 8]
        cu = (hw >> 32) * 256 + ((hw >> 8) & 0xff)              # XCC id | HW_ID[15:8]: CU, SH, SE
        ncu = int(torch.unique(cu).numel())
        span_us = float(rt1.max() - rt0.min())
        busy = float(life.sum()) / (ncu * span_us)
        # idle gaps of a CU: time during the span with NO workgroup of this launch resident
        idle = 0.0
        for c in torch.unique(cu).tolist():
            m = cu == c
            iv = sorted(zip(rt0[m].tolist(), rt1[m].tolist()))
            cur = rt0.min().item()
            for a_, b_ in iv:
                if a_ > cur:
                    idle += a_ - cur
                cur = max(cur, b_)
            idle += rt1.max().item() - cur
        names = ['setup', 'first_load', 'k_loop', 'stage', 'stats', 'copy_out', 'store_ack']
        first = rt0.sort().values
        rec = {'launch': name, 'workgroups': n, 'cus': ncu, 'wall_us': round(wall, 1), 'span_us': round(span_us, 1), 'shader_mhz': round(mhz),
               'tflops': round(flops / wall / 1e6, 1), 'mean_lifetime_us': round(float(life.mean()), 2),
               'mean_resident_per_cu': round(busy, 2), 'cu_idle_frac': round(idle / (ncu * span_us), 3),
               'phase_us': {k: round(float(d[:, i].mean()), 2) for i, k in enumerate(names)},
               'setup_split_us': {'to_tile_known': round(float(((r[:, 11].double() - t[:, 0]) / mhz).mean()), 2),
                                  'rows_decomposed': round(float(((r[:, 12].double() - r[:, 11].double()) / mhz).mean()), 2),
                                  'walker_ready': round(float(((r[:, 13].double() - r[:, 12].double()) / mhz).mean()), 2),
                                  'bias_and_dma_issue': round(float(((t[:, 1] - r[:, 13].double()) / mhz).mean()), 2)},
               'phase_p90_us': {k: round(float(d[:, i].quantile(0.9)), 2) for i, k in enumerate(names)},
               'first_round_entry_spread_us': round(float(first[min(n, 2 * ncu) - 1] - first[0]), 2)}
        print(json.dumps(rec), flush=True)

    M = 50432
    x768 = torch.randn(M, 768, device='cuda').to(bf)
    x3072 = torch.randn(M, 3072, device='cuda').to(bf)
    for (K, N, tag) in ((768, 768, 'proj'), (768, 2304, 'qkv'), (768, 3072, 'fc1'), (3072, 768, 'fc2')):
        w = (torch.randn(N, K, device='cuda') * 0.03).to(bf)
        b = torch.randn(N, device='cuda')
        y = torch.empty(M, N, device='cuda', dtype=bf)
        xin = x768 if K == 768 else x3072
        measure(f'linear {tag} M={M} K={K} N={N} fwd', lambda: check(L.saicv_linear_fwd(0, ptr(xin), ptr(w), ptr(b), ptr(y), M, K, N, 0, 0, 0, 1, st)), 2.0 * M * K * N)
    batch = 256
    for (ci, co, k, s, h) in ((64, 256, 1, 1, 56), (256, 64, 1, 1, 56), (64, 64, 3, 1, 56), (128, 512, 1, 1, 28), (128, 128, 3, 1, 28),
                              (256, 1024, 1, 1, 14), (1024, 256, 1, 1, 14), (256, 256, 3, 1, 14), (512, 2048, 1, 1, 7), (512, 512, 3, 1, 7)):
        d = ops._desc(batch, h, h, ci, co, k, k, s, k // 2, bf)
        x = torch.randn(batch, h, h, ci, device='cuda').to(bf)
        wf = (torch.randn(co, k, k, ci, device='cuda') * 0.05).to(bf)
        wd = (torch.randn(ci, k, k, co, device='cuda') * 0.05).to(bf)
        y = torch.empty(batch, d.OH, d.OW, co, device='cuda', dtype=bf)
        dy = torch.randn(batch, d.OH, d.OW, co, device='cuda').to(bf)
        dx = torch.empty_like(x)
        rows = L.saicv_conv2d_stat_rows(ctypes.byref(d))
        stats = torch.zeros(2, rows, co, device='cuda')
        flops = 2.0 * batch * d.OH * d.OW * co * k * k * ci
        measure(f'conv {ci}->{co} k{k} @{h} fwd+stats', lambda: check(L.saicv_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(wf), 0, ptr(y), 0, ptr(stats[0]), ptr(stats[1]), st)), flops)
        measure(f'conv {ci}->{co} k{k} @{h} dgrad', lambda: check(L.saicv_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dx), st)), flops)


def show(path):
    for line in open(path):
        d = json.loads(line)
        if 'error' in d:
            print(d)
            continue
        ph = d['phase_us']
        print(f"{d['launch']:44s} wall {d['wall_us']:7.1f} TF {d['tflops']:6.1f} life {d['mean_lifetime_us']:6.2f} res/cu {d['mean_resident_per_cu']:4.2f} | "
              + ' '.join(f'{k} {v:5.2f}' for k, v in ph.items()) + ' || setup: ' + ' '.join(f'{k} {v:4.2f}' for k, v in d.get('setup_split_us', {}).items()))


if __name__ == '__main__':
    if '--build' in sys.argv:
        build()
    elif '--show' in sys.argv:
        show(sys.argv[sys.argv.index('--show') + 1])
    else:
        main()
