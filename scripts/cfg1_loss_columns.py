"""BASELINE.json configs[0], the same experiment on both sides: the loss column of the REFERENCE loop on the host
(profiles/r05_cfg1_reference_cpu_epoch.log, oracle/run_reference_cifar_epoch.py) next to the engine's entry script on one MI355X
(profiles/r05_cfg1_engine_gpu_epoch.log, scripts/gpu_cfg1_r05.sh).    python scripts/cfg1_loss_columns.py > profiles/r05_cfg1_loss_columns.md"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def column(path):
    losses, tail = {}, []
    for line in open(path):
        m = re.search(r'iter \[(\d+), \d+\].*loss: ([0-9.]+)', line)
        if m:
            losses[int(m.group(1))] = float(m.group(2))
        if 'train_loss' in line or 'images/s' in line or 'acc1' in line:
            tail.append(line.strip()[:230])
    return losses, tail


ref, ref_tail = column(os.path.join(ROOT, 'profiles', 'r05_cfg1_reference_cpu_epoch.log'))
eng, eng_tail = column(os.path.join(ROOT, 'profiles', 'r05_cfg1_engine_gpu_epoch.log'))
print('# cfg-1 (ResNet18-CIFAR, batch 64, 781 iterations, SGD 0.1 / 0.9 / 5e-4): reference loop on the host vs the engine on MI355X\n')
print('Same pickle bytes (`scripts/cifar_synthetic_pickles.py`: 50 000 synthetic images with class-dependent means), same seed-0 initial')
print('weights, same `DistributedSampler(shuffle=True)` order at epoch 1, fp32 on both sides (`SAICV_CIFAR_AMP=0`), `print_interval` 10.')
print('The loop prints the loss OF THAT ITERATION.  The task is learnt within 60 iterations; during the steep descent (iterations 20-60) two')
print('correct fp32 implementations separate (SGD at lr 0.1 amplifies rounding), afterwards both sit on the same plateau.\n')
print('| iteration | reference (host, fp32) | engine (MI355X, fp32) |')
print('|---|---|---|')
for k in sorted(ref):
    if k <= 150 or k % 50 == 0 or k >= 750:
        print(f'| {k} | {ref[k]:.4f} | {eng.get(k, float("nan")):.4f} |')
both = [k for k in ref if k in eng and k >= 100]
print(f'\nIterations 100-780 ({len(both)} printed values): mean loss reference {sum(ref[k] for k in both) / len(both):.5f}, '
      f'engine {sum(eng[k] for k in both) / len(both):.5f}; largest value reference {max(ref[k] for k in both):.4f}, engine {max(eng[k] for k in both):.4f}.\n')
print('Reference: `' + ' | '.join(ref_tail[-1:]) + '`\n')
print('Engine:\n')
for t in eng_tail[-4:]:
    print('    ' + t)
