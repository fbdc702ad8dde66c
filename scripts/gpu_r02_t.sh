#!/bin/bash
# r02: DDP tests after routing every step collective through the library communicator; CIFAR config[0] epoch through the entry script
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02t
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_train_loop.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
# BASELINE.json configs[0]: ResNet18Cifar, batch 128, one epoch of 50 000 synthetic 32x32 samples through the reference-shaped
# entry script (torch.distributed.run, one rank), evaluation included
cd $GRAFT_REPO_ROOT/00.classification_training/cifar100/resnet18cifar
rm -rf checkpoints log
PYTHONPATH=$GRAFT_REPO_ROOT SAICV_CIFAR_EPOCHS=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  -m simpleaicv_pytorch_training_examples_amd.tools.train_classification_model --work-dir ./ > $O/cifar_epoch.log 2>&1
echo "rc=$?"; tail -6 $O/cifar_epoch.log | cut -c1-220
ls checkpoints 2>/dev/null | head
rm -rf checkpoints log
