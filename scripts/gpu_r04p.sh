#!/bin/bash
# BASELINE.json configs[0] through the engine's entry script on one MI355X: one full epoch (781 iterations of batch 64, 50 000
# synthetic images) + the test pass -- the GPU-side companion of profiles/r04_cfg1_reference_cpu_epoch.log
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04p; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
cd "$GRAFT_REPO_ROOT/00.classification_training/cifar100/resnet18cifar" && rm -rf checkpoints log
SAICV_CIFAR_BATCH=64 SAICV_CIFAR_EPOCHS=1 SAICV_CIFAR_WORKERS=8 SAICV_CIFAR_TEST=2048 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 -m simpleaicv_pytorch_training_examples_amd.tools.train_classification_model --work-dir ./ > $O/cfg1_engine_gpu_epoch.log 2>&1
echo "rc=$?"; grep -v "^W2026\|^$" $O/cfg1_engine_gpu_epoch.log | tail -8 | cut -c1-220
rm -rf checkpoints log
