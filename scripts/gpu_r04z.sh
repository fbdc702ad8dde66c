#!/bin/bash
# compute_macs_and_params test + the classification test entry script end to end (short synthetic val set)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 60 python -m pytest tests/test_gpu_train_loop.py -q -s -k compute_macs > $O/pytest_macs.log 2>&1; tail -3 $O/pytest_macs.log
( cd $GRAFT_REPO_ROOT/00.classification_training/imagenet/resnet50 && rm -rf log
  SAICV_CLS_TEST=1024 SAICV_CLS_WORKERS=4 timeout 60 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 -m simpleaicv_pytorch_training_examples_amd.tools.test_classification_model --work-dir ./ > $O/entry_test_classification.log 2>&1
  echo "test entry rc=$?"; grep -E "model:|acc1:" $O/entry_test_classification.log | cut -c1-200; rm -rf log )
