"""Benchmark copy of reference 00.classification_training/cifar100/resnet18cifar/train_config.py (:22-110): training
attributes as the reference sets them, CIFAR-100 pickles + transform block replaced by a synthetic dataset with the
CIFAR-100 shape (50 000 train / 10 000 test samples of 32x32x3, 100 classes).  BASELINE.json configs[0]: the
plumbing configuration (runs on CPU / gloo through tools/train_classification_model.py as well as on MI355X).
SAICV_CIFAR_PICKLES=<dir> (r05): read CIFAR-100-format pickles through CIFAR100Dataset + the config's mean / std normalisation
instead (scripts/cifar_synthetic_pickles.py writes the same bytes the reference's host run reads); SAICV_CIFAR_AMP=0: fp32."""
import os
import sys

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.append(BASE_DIR)

from SimpleAICV.classification import backbones
from SimpleAICV.classification import losses
from SimpleAICV.classification.datasets.syntheticdataset import SyntheticClassificationDataset
from SimpleAICV.classification.common import ClassificationCollater, load_state_dict


class config:
    network = 'resnet18cifar'
    num_classes = 100
    input_image_size = 32

    if os.environ.get('SAICV_CIFAR_PICKLES'):
        import torch
        torch.manual_seed(0)      # the weights oracle/run_reference_cifar_epoch.py starts from
    model = backbones.__dict__[network](**{'num_classes': num_classes})

    trained_model_path = ''
    load_state_dict(trained_model_path, model)

    train_criterion = losses.__dict__['CELoss']()
    test_criterion = losses.__dict__['CELoss']()

    _pickles = os.environ.get('SAICV_CIFAR_PICKLES')
    if _pickles:
        from SimpleAICV.classification.datasets.cifar100dataset import CIFAR100Dataset
        sys.path.append(os.path.join(BASE_DIR, 'scripts'))
        from cifar_synthetic_pickles import Normalize
        train_dataset = CIFAR100Dataset(root_dir=_pickles, set_name='train', transform=Normalize())
        test_dataset = CIFAR100Dataset(root_dir=_pickles, set_name='test', transform=Normalize())
    else:
        train_dataset = SyntheticClassificationDataset(int(os.environ.get('SAICV_CIFAR_TRAIN', 50000)), input_image_size,
                                                       num_classes, seed=0)
        test_dataset = SyntheticClassificationDataset(int(os.environ.get('SAICV_CIFAR_TEST', 10000)), input_image_size,
                                                      num_classes, seed=1)
    train_collater = ClassificationCollater()
    test_collater = ClassificationCollater()

    seed = 0
    batch_size = int(os.environ.get('SAICV_CIFAR_BATCH', 128))     # BASELINE.json configs[0] quotes batch 64
    num_workers = int(os.environ.get('SAICV_CIFAR_WORKERS', 16))
    accumulation_steps = 1

    optimizer = ('SGD', {'lr': 0.1, 'momentum': 0.9, 'global_weight_decay': False, 'weight_decay': 5e-4,
                         'no_weight_decay_layer_name_list': []})
    scheduler = ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.2, 'milestones': [60, 120, 160]})

    epochs = int(os.environ.get('SAICV_CIFAR_EPOCHS', 200))
    print_interval = int(os.environ.get('SAICV_CIFAR_PRINT', 50))

    sync_bn = False
    use_amp = os.environ.get('SAICV_CIFAR_AMP', '1') != '0'
    use_compile = False
    compile_params = {'mode': 'default'}

    use_ema_model = False
    ema_model_decay = 0.9999
