"""Sums a rocprofv3 kernel_stats.csv into kernel families: ms per step and calls per step."""
import collections
import csv
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2])
FAM = r'(igemm_nt_kernel\w*?Li(\d)ELb\dELb\d|igemm_nt|igemm_tn|bn_bwd_reduce|bn_bwd_apply|bn_act_fwd|bn_finalize_bwd|bn_finalize_fwd|bn_reduce_partials|pack_weight|maxpool|pack_input|copyBuffer|fillBuffer|sgd|adamw|avgpool|attention|sa_fwd|sa_bwd|layernorm|gelu|colsum|colreduce|row_scale)'
cat, calls = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(path)):
    n = r['Name']
    m = re.search(FAM, n)
    if m and m.group(1).startswith('igemm_nt_kernel'):
        k = 'igemm_nt mode%s' % m.group(2)
    else:
        k = m.group(1) if m else ('at::native' if 'at::native' in n else n[:40])
    cat[k] += int(r['TotalDurationNs'])
    calls[k] += int(r['Calls'])
tot = sum(cat.values())
for k, v in cat.most_common(16):
    print(f'{k:24s} {v / steps / 1e6:8.3f} ms/step {100 * v / tot:5.1f}%  {calls[k] / steps:7.1f} calls/step')
print(f'{"total":24s} {tot / steps / 1e6:8.3f} ms/step')
