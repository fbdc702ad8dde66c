"""Summary of a forced-sync (SAICV_DDP_FORCE_SYNC=1, world of one) captured-step kernel trace: where the RCCL all-reduce kernels sit
in each replayed step relative to the backward kernels (first / last weight-gradient kernel) and how many there are per step."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
opt = [i for i, r in enumerate(rows) if 'sgd_flat' in r['name'] or 'adamw_flat' in r['name']]
print(f'{len(rows)} dispatches, {len(opt)} optimizer steps in the trace')
for k in range(max(0, len(opt) - 3), len(opt) - 0):
    a = opt[k - 1] + 1 if k > 0 else 0
    step = rows[a:opt[k] + 1]
    rccl = [j for j, r in enumerate(step) if 'nccl' in r['name'].lower() or 'rccl' in r['name'].lower()]
    tn = [j for j, r in enumerate(step) if 'igemm_tn' in r['name']]
    if not rccl:
        print(f'step ending at dispatch {opt[k]}: {len(step)} kernels, NO RCCL kernels')
        continue
    inside = sum(1 for j in rccl if tn and tn[0] < j < tn[-1])
    names = sorted({step[j]['name'][:60] for j in rccl})
    print(f'step ending at dispatch {opt[k]}: {len(step)} kernels, {len(rccl)} RCCL kernels ({names}), first at position {rccl[0]}, '
          f'last at {rccl[-1]}; weight-gradient kernels span {tn[0] if tn else None}..{tn[-1] if tn else None}; '
          f'{inside} RCCL kernels start between the first and the last weight-gradient kernel of the backward pass; '
          f'RCCL time {sum(float(step[j]["dur_us"]) for j in rccl):.0f} us')
