"""Fabric traffic per kernel family from the TCC request counters of `scripts/gpu_r06.sh tcc:<model>` (scripts/pmc_fold.py output):
reads = TCC_EA0_RDREQ x 64 B (32-byte requests counted at 32 B), writes = TCC_EA0_WRREQ x 64 B -- the reading VERDICT r05 asks the
bench line to carry for ViT-B instead of the FETCH_SIZE x 2 rule, which over-counts the 64-byte K-slice requests of the LDS-DMA loads.

    python scripts/make_tcc_traffic.py <tcc json> <pmc summary json (launches per step)> <out json> [model]
(the counter CSVs hold several rows per dispatch, so the fold's row count is not a launch count: launches per step come from the
FETCH_SIZE / WRITE_SIZE summary of the same command, scripts/make_pmc_summary.py)"""
import json
import sys

from make_pmc_summary import family


def main():
    src, pmc_path, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    per_step = {k: v['launches_per_step'] for k, v in json.load(open(pmc_path))['kernels'].items()}
    model = sys.argv[4] if len(sys.argv) > 4 else 'resnet50'
    fam = {}
    for name, e in json.load(open(src)).items():
        f = family(name.replace(' ', ''))
        if f is None or 'TCC_EA0_RDREQ_sum' not in e:
            continue
        n = e['launches']
        rd32 = e.get('TCC_EA0_RDREQ_32B_sum', 0.0)
        rd = ((e['TCC_EA0_RDREQ_sum'] - rd32) * 64 + rd32 * 32) * n          # (pmc_fold stores per-launch averages)
        wr = e.get('TCC_EA0_WRREQ_sum', 0.0) * 64 * n
        a = fam.setdefault(f, {'launches': 0, 'read': 0.0, 'write': 0.0})
        a['launches'] += n
        a['read'] += rd
        a['write'] += wr
    out = {'_doc': f'fabric bytes from TCC_EA0_RDREQ / TCC_EA0_WRREQ (x 64 B; 32-byte reads at 32 B) over the eager step of `bench.py --model {model}` '
                   '(b256 bf16), rocprofv3 --pmc in a pass of its own; launches per step from ' + pmc_path, 'kernels': {}}
    for f, a in sorted(fam.items()):
        n = max(a['launches'], 1)
        lps = per_step.get(f)
        out['kernels'][f] = {'launches_per_step': lps, 'read_GB_per_step': round(a['read'] / n * lps / 1e9, 2) if lps else None,
                             'write_GB_per_step': round(a['write'] / n * lps / 1e9, 2) if lps else None,
                             'bytes_per_launch': int((a['read'] + a['write']) / n)}
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps(out['kernels'], indent=1))


if __name__ == '__main__':
    main()
