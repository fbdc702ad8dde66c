#!/bin/bash
# Round-4 GPU runner: one parametrised script instead of one file per experiment.
#   gpurun --timeout N -- 'bash scripts/gpu_r04.sh <tag> <steps...>'
# steps: tests[:<pytest args>]  testsall  testsx[:<workers>]  smoke  sweep[:linear|conv]  kbench  lbench  bench[:<bench.py args>]  prof:<model>
#        entries (the three entry scripts end to end through torch.distributed.run, DETR resumed from latest.pth)
# Companions: gpu_ab_env.sh (same-box A/B of environment switches on a model bench), gpu_lb_env.sh (the same with the
# isolated linear GEMMs), gpu_pmc_r03.sh (PMC HBM traffic of the default command), gpu_tiles.sh (tile-geometry sweep).
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/r04$TAG
mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case $name in
    tests)  timeout 1500 python -m pytest tests -m gpu -q -x ${arg:-} > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log ;;
    testsall) timeout 1800 python -m pytest tests -m gpu -q ${arg:-} > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log ;;
    testsx) timeout 900 python -m pytest tests -m gpu -q -n ${arg:-4} --dist loadfile > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log ;;   # the suite over xdist workers sharing the GPU (files stay on one worker)
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    sweep)  timeout 900 python scripts/nt_sweep.py ${arg:-all} > $O/nt_sweep_${arg:-all}.jsonl 2> $O/nt_sweep.err; tail -3 $O/nt_sweep_${arg:-all}.jsonl | cut -c1-600; tail -3 $O/nt_sweep.err ;;
    kbench) KB_ITERS=10 timeout 600 python scripts/kernel_bench.py > $O/kernel_microbench.jsonl 2> $O/kbench.err; tail -4 $O/kernel_microbench.jsonl | cut -c1-400 ;;
    lbench) timeout 600 python scripts/linear_bench.py > $O/linear_bench.jsonl 2> $O/lbench.err; cat $O/linear_bench.jsonl | cut -c1-300 ;;
    bench)  timeout 900 python bench.py ${arg:-} > $O/bench_$(echo "${arg:-default}" | tr -c 'a-zA-Z0-9\n' '_').log 2>&1; tail -1 $O/bench_$(echo "${arg:-default}" | tr -c 'a-zA-Z0-9\n' '_').log | cut -c1-700 ;;
    prof)   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$arg -o $arg -- python $GRAFT_REPO_ROOT/bench.py --model $arg --no-secondary --no-cpu-baseline --max-windows 2 --steps 5 --warmup 5 > $O/prof_$arg.log 2>&1); f=$(find $O/prof_$arg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${arg}_rocprofv3_kernel_stats.csv && head -12 $f | cut -c1-200; t=$(find $O/prof_$arg -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python scripts/trace_compact.py $t > $O/${arg}_kernel_trace_compact.csv ;;
    ddpsync)
      # world of one, the gradient synchronisation forced on (SAICV_DDP_FORCE_SYNC=1), the step captured: the RCCL kernels of the bucket
      # all-reduces must show up once per bucket per replayed step, between the backward kernels
      (cd /tmp && SAICV_DDP_FORCE_SYNC=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ddpsync -o ddpsync -- python $GRAFT_REPO_ROOT/bench.py --model ${arg:-resnet50} --no-secondary --no-cpu-baseline --no-kernel-timer --max-windows 1 --steps 5 --warmup 5 > $O/prof_ddpsync.log 2>&1)
      t=$(find $O/prof_ddpsync -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python scripts/trace_compact.py $t > $O/ddpsync_kernel_trace_compact.csv && python scripts/ddp_trace_summary.py $O/ddpsync_kernel_trace_compact.csv | tee $O/ddpsync_summary.txt ;;
    entries)
      export PYTHONPATH=$GRAFT_REPO_ROOT
      run() { timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 -m simpleaicv_pytorch_training_examples_amd.tools.$2 --work-dir ./ ; }
      ( cd $GRAFT_REPO_ROOT/03.detection_training/coco/res50_detr_yoloresize1024 && rm -rf checkpoints log
        export SAICV_DET_TRAIN=48 SAICV_DET_TEST=8 SAICV_DET_BATCH=8 SAICV_DET_WORKERS=2 SAICV_DET_EPOCHS=1 SAICV_DET_PRINT=2
        run 29521 train_detection_model > $O/entry_detr_epoch1.log 2>&1; echo "detr run 1 rc=$? $(tail -1 $O/entry_detr_epoch1.log | cut -c1-150)"
        export SAICV_DET_EPOCHS=2
        run 29522 train_detection_model > $O/entry_detr_epoch2.log 2>&1; echo "detr run 2 rc=$? $(grep -i resuming $O/entry_detr_epoch2.log | cut -c1-160)"
        rm -rf checkpoints log )
      ( cd "$GRAFT_REPO_ROOT/13.interactive_segmentation_training/13.1.sam_segmentation_training/sam_b_training" && rm -rf checkpoints log
        export SAICV_SAM_TRAIN=16 SAICV_SAM_BATCH=4 SAICV_SAM_WORKERS=2 SAICV_SAM_EPOCHS=1
        run 29523 train_interactive_segmentation_model > $O/entry_sam_epoch1.log 2>&1; echo "sam rc=$? $(tail -1 $O/entry_sam_epoch1.log | cut -c1-150)"
        rm -rf checkpoints log )
      ( cd "$GRAFT_REPO_ROOT/00.classification_training/cifar100/resnet18cifar" 2>/dev/null && rm -rf checkpoints log
        SAICV_CIFAR_TRAIN=4096 SAICV_CIFAR_TEST=512 SAICV_CIFAR_WORKERS=2 SAICV_CIFAR_EPOCHS=1 run 29524 train_classification_model > $O/entry_cifar_epoch1.log 2>&1; echo "cifar rc=$? $(tail -1 $O/entry_cifar_epoch1.log | cut -c1-150)"
        rm -rf checkpoints log ) ;;
    *) echo "unknown step $step" ;;
  esac
done
