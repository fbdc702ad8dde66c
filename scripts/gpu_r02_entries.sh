#!/bin/bash
# r02: the detection and interactive-segmentation ENTRY scripts end to end on one GPU through torch.distributed.run:
# reference configs (shortened synthetic datasets), one epoch, checkpoint written; DETR is started a second time and must
# resume from latest.pth (optimizer / scheduler state through the torch.optim layout).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02entries
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
run() { timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 -m simpleaicv_pytorch_training_examples_amd.tools.$2 --work-dir ./ ; }
cd $GRAFT_REPO_ROOT/03.detection_training/coco/res50_detr_yoloresize1024 && rm -rf checkpoints log
export SAICV_DET_TRAIN=48 SAICV_DET_TEST=8 SAICV_DET_BATCH=8 SAICV_DET_WORKERS=2 SAICV_DET_EPOCHS=1 SAICV_DET_PRINT=2
run 29521 train_detection_model > $O/detr_epoch1.log 2>&1; echo "detr run 1 rc=$? $(grep -c 'train: epoch' $O/detr_epoch1.log) log lines; $(tail -1 $O/detr_epoch1.log | cut -c1-150)"
export SAICV_DET_EPOCHS=2
run 29522 train_detection_model > $O/detr_epoch2.log 2>&1; echo "detr run 2 rc=$? $(grep -i 'resuming' $O/detr_epoch2.log | cut -c1-160)"; tail -1 $O/detr_epoch2.log | cut -c1-150
ls checkpoints; rm -rf checkpoints log
cd "$GRAFT_REPO_ROOT/13.interactive_segmentation_training/13.1.sam_segmentation_training/sam_b_training" && rm -rf checkpoints log
export SAICV_SAM_TRAIN=16 SAICV_SAM_BATCH=4 SAICV_SAM_WORKERS=2 SAICV_SAM_EPOCHS=1
run 29523 train_interactive_segmentation_model > $O/sam_epoch1.log 2>&1; echo "sam rc=$? $(tail -1 $O/sam_epoch1.log | cut -c1-150)"
ls checkpoints; rm -rf checkpoints log
grep -n "Error\|Traceback" $O/*.log | head -10
