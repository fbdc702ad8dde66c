#!/bin/bash
# MAE pre-training: trajectory tests + the entry script end to end (one short epoch, then a resumed second one)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_loop.py -m gpu -q -k "mae" -s > $O/pytest_mae.log 2>&1; grep -v "^W2026" $O/pytest_mae.log | tail -8 | cut -c1-300
export PYTHONPATH=$GRAFT_REPO_ROOT
cd "$GRAFT_REPO_ROOT/02.masked_image_modeling_training/imagenet/mae_vit_base_patch16_224" && rm -rf checkpoints log
run() { timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 -m simpleaicv_pytorch_training_examples_amd.tools.train_mae_self_supervised_model --work-dir ./ ; }
export SAICV_MAE_TRAIN=4096 SAICV_MAE_BATCH=256 SAICV_MAE_WORKERS=8 SAICV_MAE_EPOCHS=1 SAICV_MAE_PRINT=4
run 29541 > $O/entry_mae_epoch1.log 2>&1; echo "mae run 1 rc=$? $(grep -v '^W2026' $O/entry_mae_epoch1.log | tail -1 | cut -c1-200)"
export SAICV_MAE_EPOCHS=2
run 29542 > $O/entry_mae_epoch2.log 2>&1; echo "mae run 2 rc=$? $(grep -i resuming $O/entry_mae_epoch2.log | cut -c1-200)"
grep "train: epoch\|until epoch\|train done" $O/entry_mae_epoch1.log $O/entry_mae_epoch2.log | cut -c1-220 | tail -14
ls checkpoints
rm -rf checkpoints log
