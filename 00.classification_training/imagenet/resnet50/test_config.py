"""Benchmark copy of reference 00.classification_training/imagenet/resnet50/test_config.py (:20-58): network, input size, CELoss,
global batch 256 as the reference sets them; the ILSVRC2012 val set + PIL / torchvision transform block is replaced by the synthetic
dataset of the same sample contract (SAICV_CLS_TEST shortens a smoke run of tools/test_classification_model.py)."""
import os
import sys

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.append(BASE_DIR)

from SimpleAICV.classification import backbones
from SimpleAICV.classification import losses
from SimpleAICV.classification.datasets.syntheticdataset import SyntheticClassificationDataset
from SimpleAICV.classification.common import ClassificationCollater, load_state_dict


class config:
    network = 'resnet50'
    num_classes = 1000
    input_image_size = 224
    scale = 256 / 224

    model = backbones.__dict__[network](**{'num_classes': num_classes})

    trained_model_path = os.environ.get('SAICV_CLS_WEIGHTS', '')
    load_state_dict(trained_model_path, model)

    test_criterion = losses.__dict__['CELoss']()
    test_dataset = SyntheticClassificationDataset(int(os.environ.get('SAICV_CLS_TEST', 50000)), input_image_size, num_classes, seed=1)
    test_collater = ClassificationCollater()

    seed = 0
    batch_size = int(os.environ.get('SAICV_CLS_BATCH', 256))
    num_workers = int(os.environ.get('SAICV_CLS_WORKERS', 16))
