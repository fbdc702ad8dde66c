#!/bin/bash
# Copies the round-6 final measurements from gpurun_out/r06<tag>/ into profiles/ (tracked) under round-6 names.
#   bash scripts/collect_r06_profiles.sh <tag of the gpu_r06.sh run>
T=${1:-fin}
O=gpurun_out/r06$T
P=profiles
set -x
[ -f $O/bench_default.log ] && tail -1 $O/bench_default.log > $P/r06_bench_default.json
for m in resnet50 vit_base_patch16 resnet50_detr_config; do
  [ -f $O/${m}_rocprofv3_kernel_stats.csv ] && cp $O/${m}_rocprofv3_kernel_stats.csv $P/r06_${m}_rocprofv3_kernel_stats.csv
  [ -f $O/${m}_kernel_trace_compact.csv ] && python - "$O/${m}_kernel_trace_compact.csv" "$P/r06_${m}_kernel_trace_compact.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ends = [i for i, r in enumerate(rows) if r['name'].startswith(('sgd_flat', 'adamw_flat'))]
k = len(ends) // 2
step = rows[ends[k - 1] + 1:ends[k] + 1] if len(ends) > 2 else rows
w = csv.DictWriter(open(sys.argv[2], 'w', newline=''), fieldnames=list(rows[0].keys()))
w.writeheader()
w.writerows(step)          # ONE replayed step
PY
done
[ -f $P/r06_resnet50_kernel_trace_compact.csv ] && python scripts/r50_label_trace.py $P/r06_resnet50_kernel_trace_compact.csv > $P/r06_resnet50_layer_table.txt
[ -f $O/r06_pmc_hbm_traffic.json ] && cp $O/r06_pmc_hbm_traffic.json $P/
[ -f $O/r06_pmc_hbm_traffic_vit_base_patch16.json ] && cp $O/r06_pmc_hbm_traffic_vit_base_patch16.json $P/
for m in resnet50 vit_base_patch16; do
  if [ -f $O/tcc_$m.json ]; then
    cp $O/tcc_$m.json $P/r06_tcc_counters_$m.json
    s=""; [ $m != resnet50 ] && s="_$m"
    (cd scripts && python make_tcc_traffic.py ../$O/tcc_$m.json ../$P/r06_pmc_hbm_traffic$s.json ../$P/r06_tcc_traffic$s.json $m > /dev/null)
  fi
done
[ -f $O/smoke.log ] && cp $O/smoke.log $P/r06_smoke.log
ls -la $P | grep r06_
