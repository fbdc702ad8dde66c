"""Benchmark copy of reference 00.classification_training/imagenet/vit_base_patch16_for_self_train_mae_pretrain/
train_config.py (:21-138): training attributes as the reference sets them (drop-path 0.1, global pool, soft-label
loss, Mixup / CutMix collater with label smoothing, AdamW with layer-wise lr decay 0.65, cosine schedule with 5 warm-up
epochs); the ILSVRC2012 dataset + OpenCV / torchvision / RandAugment transform block is replaced by a synthetic
dataset, and no MAE-pretrained checkpoint is loaded (none exists in the bench image).  BASELINE.json configs[2]."""
import os
import sys

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.append(BASE_DIR)

from SimpleAICV.classification import backbones
from SimpleAICV.classification import losses
from SimpleAICV.classification.datasets.syntheticdataset import SyntheticClassificationDataset
from SimpleAICV.classification.common import ClassificationCollater, MixupCutmixClassificationCollater, load_state_dict


class config:
    network = 'vit_base_patch16'
    num_classes = 1000
    input_image_size = 224
    scale = 256 / 224

    model = backbones.__dict__[network](**{
        'image_size': 224,
        'drop_path_prob': 0.1,
        'global_pool': True,
        'num_classes': num_classes,
    })

    trained_model_path = ''
    load_state_dict(trained_model_path, model, loading_new_input_size_position_encoding_weight=True)

    train_criterion = losses.__dict__['OneHotLabelCELoss']()
    test_criterion = losses.__dict__['CELoss']()

    train_dataset = SyntheticClassificationDataset(1281167, input_image_size, num_classes, seed=0)
    test_dataset = SyntheticClassificationDataset(50000, input_image_size, num_classes, seed=1)
    train_collater = MixupCutmixClassificationCollater(use_mixup=True, mixup_alpha=0.8, cutmix_alpha=1.0,
                                                       cutmix_minmax=None, mixup_cutmix_prob=1.0,
                                                       switch_to_cutmix_prob=0.5, mode='batch', correct_lam=True,
                                                       label_smoothing=0.1, num_classes=1000)
    test_collater = ClassificationCollater()

    seed = 0
    batch_size = 256        # total over all GPUs (the bench overrides it with per-GPU batch x GPUs: weak scaling)
    num_workers = 32
    accumulation_steps = 1

    optimizer = ('AdamW', {'lr': 5e-4, 'global_weight_decay': False, 'weight_decay': 5e-2, 'lr_layer_decay': 0.65,
                           'lr_layer_decay_block': model.blocks, 'block_name': 'blocks',
                           'no_weight_decay_layer_name_list': ['position_encoding', 'cls_token']})
    scheduler = ('CosineLR', {'warm_up_epochs': 5, 'min_lr': 1e-6})

    epochs = 100
    print_interval = 100

    sync_bn = False
    use_amp = True
    use_compile = False
    compile_params = {'mode': 'default'}

    use_ema_model = False
    ema_model_decay = 0.9999
