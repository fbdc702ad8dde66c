#!/bin/bash
# r02 evidence set: the GPU test suite, smoke(), the default bench line, rocprofv3 kernel stats of the bench workloads,
# the two PMC passes for HBM traffic.  Summaries are copied into profiles/ by hand afterwards.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02final
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-400
for m in "sam_b --batch 20 --steps 3 --warmup 2" "sam_b_encoder --batch 20 --steps 4 --warmup 2" "resnet50_detr_config --batch 8 --steps 5 --warmup 3" "resnet50_detr --batch 8 --steps 5 --warmup 3"; do
  n=$(echo $m | cut -d' ' -f1)
  timeout 900 python bench.py --model $m --no-cpu-baseline --no-secondary --max-windows 3 > $O/bench_$n.log 2>&1; echo "$n: $(tail -1 $O/bench_$n.log | cut -c1-200)"
done
B="--no-cpu-baseline --no-secondary --max-windows 1 --no-kernel-timer"
cd /tmp
prof() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python $GRAFT_REPO_ROOT/bench.py "$@" $B > $O/$name.log 2>&1; echo "$name rc=$?"; }
prof r50 --model resnet50 --steps 5 --warmup 5
prof vit --model vit_base_patch16 --steps 5 --warmup 5
prof samfull --model sam_b --batch 20 --steps 2 --warmup 1
prof detrcfg --model resnet50_detr_config --batch 8 --steps 3 --warmup 2
mkdir -p $O/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc -o $c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --eager $B > $O/pmc/$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
python scripts/make_pmc_summary.py $O/pmc 3 $O/pmc_hbm_traffic.json > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log
rm -f $O/*/*kernel_trace.csv $O/pmc/*counter_collection.csv $O/pmc/*kernel_trace.csv
for n in r50 vit samfull detrcfg; do python scripts/prof_categories.py $O/$n/${n}_kernel_stats.csv 1 | tail -1; done
