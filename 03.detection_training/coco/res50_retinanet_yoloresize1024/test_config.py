"""Benchmark copy of reference 03.detection_training/coco/res50_retinanet_yoloresize1024/test_config.py (:18-95): network,
RetinaLoss / RetinaDecoder settings, collater and evaluation switches are the train config's (the reference repeats them literally;
here they are taken from train_config.py next to this file); COCO val2017 is replaced by the synthetic detection set, batch 32 / 16
workers as the reference sets them (SAICV_DET_* shorten a smoke run of tools/test_detection_model.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from train_config import config as _train  # noqa: E402


class config:
    network, num_classes, input_image_size = _train.network, _train.num_classes, _train.input_image_size
    model = _train.model
    trained_model_path = _train.trained_model_path
    test_criterion, decoder = _train.test_criterion, _train.decoder
    test_dataset, test_collater = _train.test_dataset, _train.test_collater
    eval_type = os.environ.get('SAICV_DET_EVAL', _train.eval_type)          # 'COCO' or 'VOC'
    eval_voc_iou_threshold_list = _train.eval_voc_iou_threshold_list
    seed = 0
    batch_size = int(os.environ.get('SAICV_DET_BATCH', 32))
    num_workers = int(os.environ.get('SAICV_DET_WORKERS', 16))
