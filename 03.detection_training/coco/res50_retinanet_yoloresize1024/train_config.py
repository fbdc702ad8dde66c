"""Benchmark copy of reference 03.detection_training/coco/res50_retinanet_yoloresize1024/train_config.py (:20-162): network,
RetinaLoss / RetinaDecoder settings, collater (1024 canvas, yolo style, 100 annotation rows), global batch 32, AdamW 1e-4,
MultiStepLR [8, 12] over 13 epochs, AMP, as the reference sets them; the COCO dataset + OpenCV transform block is replaced by a
synthetic detection dataset and no pretrained backbone is loaded (neither exists in the bench image)."""
import os
import sys

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.append(BASE_DIR)

from SimpleAICV.detection import models
from SimpleAICV.detection import losses
from SimpleAICV.detection import decode
from SimpleAICV.detection.datasets.syntheticdataset import SyntheticDetectionDataset
from SimpleAICV.detection.common import DetectionCollater, load_state_dict

_PYRAMID = {'areas': [[32, 32], [64, 64], [128, 128], [256, 256], [512, 512]], 'ratios': [0.5, 1, 2],
            'scales': [2**0, 2**(1.0 / 3.0), 2**(2.0 / 3.0)], 'strides': [8, 16, 32, 64, 128]}


class config:
    network = 'resnet50_retinanet'
    num_classes = 80
    input_image_size = [1024, 1024]

    backbone_pretrained_path = ''
    model = models.__dict__[network](**{'backbone_pretrained_path': backbone_pretrained_path, 'num_classes': num_classes})

    trained_model_path = ''
    load_state_dict(trained_model_path, model)

    _loss_kwargs = dict(_PYRAMID, alpha=0.25, gamma=2, beta=1.0 / 9.0, cls_loss_weight=1., box_loss_weight=1., box_loss_type='SmoothL1')
    train_criterion = losses.__dict__['RetinaLoss'](**_loss_kwargs)
    test_criterion = losses.__dict__['RetinaLoss'](**_loss_kwargs)
    decoder = decode.__dict__['RetinaDecoder'](**dict(_PYRAMID, max_object_num=100, min_score_threshold=0.05, topn=1000,
                                                     nms_type='python_nms', nms_threshold=0.5))

    # sizes of COCO train2017 / val2017; SAICV_DET_* shorten a smoke run of the entry script
    train_dataset = SyntheticDetectionDataset(int(os.environ.get('SAICV_DET_TRAIN', 117266)), 768, 1024, num_classes=num_classes, seed=0)
    test_dataset = SyntheticDetectionDataset(int(os.environ.get('SAICV_DET_TEST', 4952)), 768, 1024, num_classes=num_classes, seed=1)
    train_collater = DetectionCollater(resize=input_image_size[0], resize_type='yolo_style', max_annots_num=100)
    test_collater = DetectionCollater(resize=input_image_size[0], resize_type='yolo_style', max_annots_num=100)

    seed = 0
    batch_size = int(os.environ.get('SAICV_DET_BATCH', 32))
    num_workers = int(os.environ.get('SAICV_DET_WORKERS', 32))
    accumulation_steps = 1

    optimizer = ('AdamW', {'lr': 1e-4, 'global_weight_decay': False, 'weight_decay': 1e-3, 'no_weight_decay_layer_name_list': []})
    scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [8, 12]})

    epochs = int(os.environ.get('SAICV_DET_EPOCHS', 13))
    print_interval = int(os.environ.get('SAICV_DET_PRINT', 100))

    eval_type = 'COCO'
    eval_epoch = [1, 3, 5, 8, 10, 12, 13]
    eval_voc_iou_threshold_list = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]
    save_model_metric = 'IoU=0.50:0.95,area=all,maxDets=100,mAP'

    sync_bn = False
    use_amp = True
    use_compile = False
    compile_params = {'mode': 'default'}

    use_ema_model = False
    ema_model_decay = 0.9999
