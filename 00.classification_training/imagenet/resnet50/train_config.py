"""Benchmark copy of reference 00.classification_training/imagenet/resnet50/train_config.py (:22-103): every training
attribute (network, criterion, batch size, optimizer, scheduler, AMP, EMA flags) as the reference sets it; the
ILSVRC2012 dataset + OpenCV / torchvision transform block is replaced by a synthetic dataset of the same sample
contract (no dataset, cv2 or torchvision in the bench image).  BASELINE.json configs[1]."""
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

    trained_model_path = ''
    load_state_dict(trained_model_path, model)

    train_criterion = losses.__dict__['CELoss']()
    test_criterion = losses.__dict__['CELoss']()

    train_dataset = SyntheticClassificationDataset(1281167, input_image_size, num_classes, seed=0)
    test_dataset = SyntheticClassificationDataset(50000, input_image_size, num_classes, seed=1)
    train_collater = ClassificationCollater()
    test_collater = ClassificationCollater()

    seed = 0
    batch_size = 256        # total over all GPUs (the bench overrides it with per-GPU batch x GPUs: weak scaling)
    num_workers = 20
    accumulation_steps = 1

    optimizer = ('SGD', {'lr': 0.1, 'momentum': 0.9, 'global_weight_decay': False, 'weight_decay': 1e-4,
                         'no_weight_decay_layer_name_list': []})
    scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [30, 60, 90]})

    epochs = 100
    print_interval = 100

    sync_bn = False
    use_amp = True
    use_compile = False
    compile_params = {'mode': 'default'}

    use_ema_model = False
    ema_model_decay = 0.9999
