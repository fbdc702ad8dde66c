#!/bin/bash
# BASELINE.json configs[0] through the engine's entry script on one MI355X, the SAME experiment as the reference's host run
# (oracle/run_reference_cifar_epoch.py -> profiles/r05_cfg1_reference_cpu_epoch.log): same pickle bytes, seed-0 weights, sampler order, fp32.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05cfg1; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
python scripts/cifar_synthetic_pickles.py /tmp/cifar_pickles > $O/pickles.log 2>&1
cd "$GRAFT_REPO_ROOT/00.classification_training/cifar100/resnet18cifar" && rm -rf checkpoints log
SAICV_CIFAR_PICKLES=/tmp/cifar_pickles SAICV_CIFAR_AMP=0 SAICV_CIFAR_PRINT=10 SAICV_CIFAR_BATCH=64 SAICV_CIFAR_EPOCHS=1 SAICV_CIFAR_WORKERS=8 \
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
  -m simpleaicv_pytorch_training_examples_amd.tools.train_classification_model --work-dir ./ > $O/cfg1_engine_gpu_epoch.log 2>&1
echo "rc=$?"; grep -v "^W2026\|^$" $O/cfg1_engine_gpu_epoch.log | grep "iter \[000[1-5]0\|iter \[007[5-8]0\|train_loss\|acc1\|Error" | cut -c1-200 | tail -16
rm -rf checkpoints log
