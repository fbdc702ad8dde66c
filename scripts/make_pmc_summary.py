"""Folds the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
HBM bytes per launch per kernel family, with the gfx950 correction MI355X_MICROARCH.md prescribes (FETCH_SIZE
reports half of wide coalesced reads -> doubled; WRITE_SIZE uncorrected; both counters are in KiB).

    python scripts/make_pmc_summary.py gpurun_out/r01d/pmc 3 profiles/r01d_pmc_hbm_traffic.json
"""
import collections
import csv
import json
import sys

FAMILIES = ['igemm_nt', 'igemm_tn', 'bn_bwd_apply', 'bn_bwd_reduce', 'bn_act_fwd', 'maxpool', 'sgd_flat', 'adamw_flat', 'layernorm_fwd',
            'layernorm_bwd', 'attention_bwd', 'attention_fwd', 'sa_fwd', 'sa_bwd', 'row_scale', 'pack_weight']


def family(name):
    if 'pw_stream' in name:          # the streaming forms of the same products (csrc/pwstream.hip) are priced with the tiled kernel's family
        return 'igemm_nt'
    for f in FAMILIES:
        if f in name:
            return f
    return None


def main():
    d, steps, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    model = sys.argv[4] if len(sys.argv) > 4 else 'resnet50'
    agg = collections.defaultdict(lambda: {'launches': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        seen = collections.Counter()
        for r in csv.DictReader(open(f'{d}/{c}_counter_collection.csv')):
            if r['Counter_Name'] != c:
                continue
            k = family(r['Kernel_Name'])
            if k is None:
                continue
            agg[k][c] += float(r['Counter_Value'])
            seen[k] += 1
        for k, n in seen.items():
            agg[k]['launches'] = max(agg[k]['launches'], n)
    out = {'_doc': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over '
                   f'`python bench.py --model {model} --steps 2 --warmup 1 --eager` (b256 bf16); counters are KiB; FETCH_SIZE doubled per '
                   'MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); WRITE_SIZE uncorrected',
           'steps_profiled': steps, 'kernels': {}}
    for k, v in sorted(agg.items()):
        rd = v['FETCH_SIZE'] * 1024 * 2
        wr = v['WRITE_SIZE'] * 1024
        n = max(v['launches'], 1)
        out['kernels'][k] = {'launches_per_step': round(n / steps, 1), 'read_GB_per_step': round(rd / steps / 1e9, 2),
                             'write_GB_per_step': round(wr / steps / 1e9, 2), 'bytes_per_launch': int((rd + wr) / n)}
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps(out['kernels'], indent=1))


if __name__ == '__main__':
    main()
