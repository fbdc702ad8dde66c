"""Per-layer speed-of-light for the ResNet-50 step at per-GPU batch 256 (bf16 operands, fp32 weight gradients):
for every convolution, forward / data-gradient / weight-gradient time is bounded below by
max(flops / MFMA peak, algorithmic bytes / HBM peak).  Prints the sum, and how it splits into layers that are
HBM-bound and MFMA-bound at those peaks, so the measured igemm time can be priced against what the SHAPES allow.
Peaks: /opt/skills/guides/MI355X_MICROARCH.md (2.5 PFLOP/s dense bf16, 8 TB/s HBM3E)."""
import sys

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
PEAK_F, PEAK_B = 2.5e15, 8.0e12


def resnet50_convs():
    out = [(3, 64, 7, 2, 224)]
    h, cin = 56, 64
    for planes, blocks, stride in [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]:
        for b in range(blocks):
            s = stride if b == 0 else 1
            out.append((cin, planes, 1, 1, h))
            out.append((planes, planes, 3, s, h))
            out.append((planes, planes * 4, 1, 1, h // s))
            if b == 0:
                out.append((cin, planes * 4, 1, s, h))
            cin, h = planes * 4, h // s
    return out


def main():
    tot = {'fwd': [0, 0, 0.0], 'dgrad': [0, 0, 0.0], 'wgrad': [0, 0, 0.0]}
    split = {'hbm': 0.0, 'mfma': 0.0}
    rows = []
    for i, (ci, co, k, s, h) in enumerate(resnet50_convs()):
        oh = h // s
        flops = 2.0 * B * oh * oh * co * ci * k * k
        xin = B * h * h * ci * 2 if not (k == 1 and s == 2) else B * oh * oh * ci * 2      # strided 1x1 reads a quarter
        yout = B * oh * oh * co * 2
        w = co * ci * k * k
        legs = {'fwd': xin + w * 2 + yout, 'dgrad': yout + w * 2 + B * h * h * ci * 2, 'wgrad': xin + yout + w * 4}
        if i == 0:
            legs.pop('dgrad')
        for leg, byts in legs.items():
            t = max(flops / PEAK_F, byts / PEAK_B)
            bound = 'hbm' if byts / PEAK_B > flops / PEAK_F else 'mfma'
            tot[leg][0] += flops
            tot[leg][1] += byts
            tot[leg][2] += t
            split[bound] += t
        rows.append((ci, co, k, s, h, flops / 1e9, legs['fwd'] / 1e6, flops / legs['fwd']))
    print(f'ResNet-50 convolutions, batch {B}: flops/byte ridge = {PEAK_F / PEAK_B:.0f}')
    for leg, (f, b, t) in tot.items():
        print(f'  {leg:6s} {f / 1e12:6.2f} TFLOP  {b / 1e9:6.2f} GB  lower bound {t * 1e3:6.3f} ms')
    print(f'  sum of lower bounds {sum(v[2] for v in tot.values()) * 1e3:.3f} ms  (HBM-bound legs {split["hbm"] * 1e3:.3f} ms, '
          f'MFMA-bound legs {split["mfma"] * 1e3:.3f} ms)')
    if '-v' in sys.argv:
        for r in rows:
            print('   %4d->%4d k%d s%d @%3d  %8.1f GFLOP  %8.1f MB  %6.0f flop/B' % r)


if __name__ == '__main__':
    main()
