"""Label the convolution launches of one ResNet-50 training step (scripts/trace_compact.py output of a rocprofv3 kernel trace) with
their layer and price each against its shape bound:   python scripts/r50_label_trace.py <compact.csv> [step index]
The eager order is fixed: forward convolutions in module order (stem; per Bottleneck conv1, downsample, conv2, conv3), backward in
reverse (after the classifier's two launches) with one data-gradient launch (igemm_nt / pw_stream; none for the stem, strided ones run one grid of parity classes) per
convolution.  Bounds: max(flops / 2.5 PFLOP/s, algorithmic bytes incl. the fused epilogue operands / 8 TB/s)."""
import csv
import sys

B = 256


def convs():
    out = [('stem', 3, 64, 7, 2, 224)]
    h, cin = 56, 64
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
        for b in range(blocks):
            s = stride if b == 0 else 1
            out.append((f'l{li}.{b}.c1', cin, planes, 1, 1, h))
            if b == 0:
                out.append((f'l{li}.{b}.ds', cin, planes * 4, 1, s, h))
            out.append((f'l{li}.{b}.c2', planes, planes, 3, s, h))
            out.append((f'l{li}.{b}.c3', planes, planes * 4, 1, 1, h // s))
            cin, h = planes * 4, h // s
    return out


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ends = [i for i, r in enumerate(rows) if r['name'].startswith('sgd_flat')]
    if len(ends) == 1:                      # a one-step file (profiles/r05_resnet50_kernel_trace_compact.csv)
        step = rows
    else:
        k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) // 2
        step = rows[ends[k - 1] + 1:ends[k] + 1]
    nt = [r for r in step if 'igemm_nt1' in r['name'] or 'pw_stream' in r['name'] or 'igemm_nt_stream' in r['name']]
    cv = convs()
    nf = len(cv)
    # (nt[nf], nt[nf + 1]: the classifier's forward and data gradient)
    fwd, bwd = nt[:nf], nt[nf + 2:nf + 2 + nf - 1]
    assert len(nt) == 2 * nf + 1, (len(nt), nf)
    tot = {'fwd': [0.0, 0.0], 'dgrad': [0.0, 0.0]}
    print(f'{"layer":10s} {"shape":22s} | fwd us  bound  frac kernel | dgrad us bound  frac kernel')
    for i, (name, ci, co, kk, s, h) in enumerate(cv):
        oh = h // s
        flops = 2.0 * B * oh * oh * co * ci * kk * kk
        xin = B * (oh * oh if (kk == 1 and s == 2) else h * h) * ci * 2
        yout = B * oh * oh * co * 2
        fb = max(flops / 2.5e15, (xin + yout) / 8e12) * 1e6
        f = fwd[i]
        line = f'{name:10s} {ci:4d}->{co:4d} k{kk} s{s} @{h:3d} | {float(f["dur_us"]):6.1f} {fb:6.1f} {fb / float(f["dur_us"]):5.2f} {"pw" if "pw_stream" in f["name"] else "nt":3s}'
        tot['fwd'][0] += float(f['dur_us']); tot['fwd'][1] += fb
        if i > 0:
            d = bwd[nf - 1 - i]
            # data gradient: reads dy, writes dx; fused: BatchNorm-backward sums read y of the producer (same size as dx) + mask,
            # c1 of a block adds the shortcut gradient (same size as dx)
            extra = B * h * h * ci * 2 * (2 if name.endswith('c1') else 1) if not name.startswith('l1.0') or True else 0
            db = max(flops / 2.5e15, (yout + B * h * h * ci * 2 + extra) / 8e12) * 1e6
            line += f' | {float(d["dur_us"]):6.1f} {db:6.1f} {db / float(d["dur_us"]):5.2f} {"pw" if "pw_stream" in d["name"] else "nt":3s}'
            tot['dgrad'][0] += float(d['dur_us']); tot['dgrad'][1] += db
        print(line)
    for k_, (t, b) in tot.items():
        print(f'{k_}: {t / 1e3:.3f} ms measured, {b / 1e3:.3f} ms bound, {b / t:.2f}')


if __name__ == '__main__':
    main()
