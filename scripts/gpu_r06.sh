#!/bin/bash
# Round-6 GPU runner.   gpurun --timeout N -- 'bash scripts/gpu_r06.sh <tag> <step> <step> ...'
# steps (":"-separated arguments; "+" inside an env list separates variables, "|" separates variants):
#   seeds                              scripts/pick_smoke_seed.py (smoke fixture over data seeds, own gates and equal gates)
#   cfg1                               BASELINE.json configs[0] through the entry script on the reference run's pickles (scripts/gpu_cfg1_r05.sh)
#   tests[:<pytest args>] / testsx[:<workers>] / smoke
#   ab:<model>:<envA>|<envB>|...       same-box A/B of environment variants on one model bench (variants interleaved, REPS rounds, default 2);
#                                      an empty variant ("-") is the default build
#   abl:<model>:<libA>|<libB>|...      same-box A/B of LIBRARY variants (scripts/build_variant_lib.py; "-" = the product library)
#   tl[:<env>]                         igemm_nt1 per-workgroup phase timeline (scripts/nt_timeline.py)
#   kab:<envA>|<envB>|...              the same with scripts/kernel_bench.py (per-shape ResNet-50 convolutions), one run each
#   lab:<envA>|<envB>|...              the same with scripts/linear_fused_bench.py (ViT-B layer GEMMs)
#   sq:<model>[:<env>]                 SQ counters (two passes) of every igemm / bn / layernorm kernel inside the eager step
#   tcc:<model>[:<env>]                L2 hit / miss + fabric read / write requests per kernel family inside the eager step
#   bench[:<bench.py args>]            one bench line (default = the driver's command)
#   prof:<model>                       rocprofv3 --kernel-trace --stats of the bench command + compact one-step trace
#   pmc:<model>                        HBM traffic (FETCH_SIZE / WRITE_SIZE passes) -> scripts/make_pmc_summary.py
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/r06$TAG
mkdir -p $O
export TMPDIR=/tmp
REPS=${REPS:-2}
benchval() { tail -1 "$1" | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); kb=d.get("kernel_breakdown_ms_per_step",{})
    print(d["ms_per_step"], d["value"], {k: round(v,3) for k,v in kb.items()} if kb else "")
except Exception as e: print("ERR", e)' 2>&1 | tail -1; }
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case $name in
    seeds)  timeout 600 python scripts/pick_smoke_seed.py 1 2 3 4 > $O/smoke_seed.txt 2> $O/smoke_seed.err; cut -c1-170 $O/smoke_seed.txt ;;
    cfg1)   bash scripts/gpu_cfg1_r05.sh ;;
    testsall) timeout 2400 python -m pytest tests -m gpu -q -s ${arg:-} > $O/pytest_gpu_all.log 2>&1; grep -E "^\[|^FAILED|^ERROR|passed|failed" $O/pytest_gpu_all.log | cut -c1-400 | tail -80 ;;
    repeat)   # repeat:<N>[:<pytest args>]  the driver's command (-x) N times on this box, one log per run + a summary line each
      n=${arg%%:*}; extra=""; [[ "$arg" == *:* ]] && extra=${arg#*:}
      for i in $(seq 1 $n); do
        timeout 1500 python -m pytest tests -m gpu -x -q $extra > $O/pytest_gpu_run$i.log 2>&1
        echo "run $i rc=$? $(tail -1 $O/pytest_gpu_run$i.log | cut -c1-200)" | tee -a $O/pytest_gpu_repeat_summary.txt
      done ;;
    probe)    # probe:<script under scripts/probes>[:<args separated by +>]
      sc=${arg%%:*}; a=""; [[ "$arg" == *:* ]] && a=$(echo "${arg#*:}" | tr '+' ' ')
      timeout 900 python scripts/probes/$sc.py $a > $O/probe_${sc}_$(echo "$a" | tr -c 'a-zA-Z0-9\n' '_').log 2>&1; tail -${PROBE_TAIL:-60} $O/probe_${sc}_$(echo "$a" | tr -c 'a-zA-Z0-9\n' '_').log | cut -c1-260 ;;
    tests)  timeout 1500 python -m pytest tests -m gpu -q ${arg:-} > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log | cut -c1-300 ;;
    testsx) timeout 900 python -m pytest tests -m gpu -q -n ${arg:-4} --dist loadfile > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    ab)
      model=${arg%%:*}; vars=${arg#*:}
      IFS='|' read -ra V <<< "$vars"
      for rep in $(seq 1 $REPS); do
        i=0
        for v in "${V[@]}"; do
          i=$((i+1)); e=$(echo "$v" | tr '+' ' '); [ "$v" == "-" ] && e=""
          env $e timeout 600 python bench.py --model $model --no-secondary --no-cpu-baseline --no-sam --max-windows ${ABWIN:-2} > $O/ab_${model}_${i}_$rep.log 2>&1
          echo "ab $model variant $i [$e] rep $rep: $(benchval $O/ab_${model}_${i}_$rep.log)"
        done
      done ;;
    abl)   # abl:<model>:<libtagA>|<libtagB>...   ("-" = the product library; variants from scripts/build_variant_lib.py)
      model=${arg%%:*}; vars=${arg#*:}
      IFS='|' read -ra V <<< "$vars"
      for rep in $(seq 1 $REPS); do
        for v in "${V[@]}"; do
          n=$(echo "$v" | tr -c 'a-zA-Z0-9\n' '_')
          timeout 600 python scripts/with_lib.py "$v" bench.py --model $model --no-secondary --no-cpu-baseline --no-sam --max-windows ${ABWIN:-2} > $O/abl_${model}_${n}_$rep.log 2>&1
          echo "abl $model lib [$v] rep $rep: $(benchval $O/abl_${model}_${n}_$rep.log)"
        done
      done ;;
    tl)    # tl[:<env>]  per-workgroup phase timeline of igemm_nt1_kernel (debug library built by `python scripts/nt_timeline.py --build`)
      e=$(echo "$arg" | tr '+' ' ')
      env $e timeout 300 python scripts/nt_timeline.py > $O/nt_timeline$(echo "$arg" | tr -c 'a-zA-Z0-9\n' '_').jsonl 2> $O/nt_timeline.err; tail -2 $O/nt_timeline.err
      python scripts/nt_timeline.py --show $O/nt_timeline$(echo "$arg" | tr -c 'a-zA-Z0-9\n' '_').jsonl ;;
    kab)
      IFS='|' read -ra V <<< "$arg"; i=0
      for v in "${V[@]}"; do
        i=$((i+1)); e=$(echo "$v" | tr '+' ' '); [ "$v" == "-" ] && e=""
        env $e KB_ITERS=${KB_ITERS:-10} timeout 600 python scripts/kernel_bench.py > $O/kab_$i.jsonl 2> $O/kab_$i.err
        echo "kab variant $i [$e]: $(tail -3 $O/kab_$i.jsonl | head -1 | cut -c1-300)"
      done ;;
    lab)
      IFS='|' read -ra V <<< "$arg"; i=0
      for v in "${V[@]}"; do
        i=$((i+1)); e=$(echo "$v" | tr '+' ' '); [ "$v" == "-" ] && e=""
        env $e timeout 600 python scripts/linear_fused_bench.py > $O/lab_$i.jsonl 2> $O/lab_$i.err
        echo "lab variant $i [$e]: $(tail -1 $O/lab_$i.jsonl | cut -c1-300)"
      done ;;
    sq|tcc)
      model=${arg%%:*}; e=""; [[ "$arg" == *:* ]] && e=$(echo "${arg#*:}" | tr '+' ' ')
      if [ $name == sq ]; then
        P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
        P2="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
      else
        P1="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
        P2="TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum"
      fi
      n=0
      for P in "$P1" "$P2"; do
        n=$((n+1))
        (cd /tmp && env $e timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/${name}_${model}_$n -o p -- python $GRAFT_REPO_ROOT/bench.py --model $model --steps 2 --warmup 1 --eager --no-cpu-baseline --no-secondary --no-sam --no-kernel-timer --max-windows 1 > $O/${name}_${model}_$n.log 2>&1)
      done
      python scripts/pmc_fold.py $O/${name}_${model}_1 $O/${name}_${model}_2 > $O/${name}_${model}.json 2> $O/${name}_${model}.err; head -c 600 $O/${name}_${model}.json; tail -2 $O/${name}_${model}.err; rm -rf $O/${name}_${model}_1 $O/${name}_${model}_2 ;;
    bench)  timeout 900 python bench.py ${arg:-} > $O/bench_$(echo "${arg:-default}" | tr -c 'a-zA-Z0-9\n' '_').log 2>&1; tail -1 $O/bench_$(echo "${arg:-default}" | tr -c 'a-zA-Z0-9\n' '_').log | cut -c1-900 ;;
    prof)   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$arg -o $arg -- python $GRAFT_REPO_ROOT/bench.py --model $arg --no-secondary --no-cpu-baseline --no-sam --max-windows 2 --steps 5 --warmup 5 > $O/prof_$arg.log 2>&1); f=$(find $O/prof_$arg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${arg}_rocprofv3_kernel_stats.csv && head -12 $f | cut -c1-200; t=$(find $O/prof_$arg -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python scripts/trace_compact.py $t > $O/${arg}_kernel_trace_compact.csv; rm -rf $O/prof_$arg ;;      # (raw traces: tens of MiB each, the merge-back limit is 64 MiB)
    pmc)    # HBM traffic: separate FETCH_SIZE / WRITE_SIZE passes of the eager step, folded per kernel family (MI355X_MICROARCH.md corrections)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$arg -o $c -- python $GRAFT_REPO_ROOT/bench.py --model $arg --steps 2 --warmup 1 --eager --no-cpu-baseline --no-secondary --no-sam --no-power --no-kernel-timer --max-windows 1 > $O/pmc_${arg}_$c.log 2>&1); echo "pmc $arg $c rc=$?"
      done
      suffix=""; [ "$arg" != "resnet50" ] && suffix="_$arg"
      python scripts/make_pmc_summary.py $O/pmc_$arg 3 $O/r06_pmc_hbm_traffic$suffix.json $arg | head -30; rm -rf $O/pmc_$arg ;;
    *) echo "unknown step $step" ;;
  esac
done
