"""Benchmark copy of reference 02.masked_image_modeling_training/imagenet/mae_vit_base_patch16_224/train_config.py
(:22-100): the training attributes as the reference sets them (mask ratio 0.75, MSE on per-patch normalised pixels,
AdamW lr 6e-4 betas (0.9, 0.95) weight decay 0.05 with 1-d parameters at 0, cosine schedule with 40 warm-up epochs,
AMP, total batch 1024); the ILSVRC2012 dataset + OpenCV / torchvision transform block is replaced by a synthetic
dataset (no dataset, OpenCV or torchvision in the bench image).  SAICV_MAE_* environment variables shrink the run for
smoke tests (scripts/gpu_r04q.sh)."""
import os
import sys

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.append(BASE_DIR)

from SimpleAICV.masked_image_modeling import models
from SimpleAICV.masked_image_modeling import losses
from SimpleAICV.classification.datasets.syntheticdataset import SyntheticClassificationDataset
from SimpleAICV.classification.common import load_state_dict
from SimpleAICV.masked_image_modeling.common import MAESelfSupervisedPretrainCollater


class config:
    network = 'vit_base_patch16_224_mae_pretrain_model'
    input_image_size = 224
    scale = 256 / 224

    model = models.__dict__[network](**{'mask_ratio': 0.75})

    trained_model_path = ''
    load_state_dict(trained_model_path, model)

    train_criterion = losses.__dict__['MSELoss']()

    train_dataset = SyntheticClassificationDataset(int(os.environ.get('SAICV_MAE_TRAIN', 1281167)), input_image_size, 1000, seed=0)
    train_collater = MAESelfSupervisedPretrainCollater(image_size=input_image_size, patch_size=16, norm_label=True)

    seed = 0
    batch_size = int(os.environ.get('SAICV_MAE_BATCH', 1024))       # total over all GPUs
    num_workers = int(os.environ.get('SAICV_MAE_WORKERS', 32))     # total over all GPUs
    accumulation_steps = 1

    # lr = 1.5e-4 * batch_size * accumulation_steps / 256; 1-d parameters (biases, norms, tokens) at weight decay 0
    optimizer = ('AdamW', {'lr': 6e-4, 'global_weight_decay': False, 'weight_decay': 5e-2, 'no_weight_decay_layer_name_list': [],
                           'beta1': 0.9, 'beta2': 0.95})
    scheduler = ('CosineLR', {'warm_up_epochs': 40, 'min_lr': 1e-6})

    epochs = int(os.environ.get('SAICV_MAE_EPOCHS', 400))
    print_interval = int(os.environ.get('SAICV_MAE_PRINT', 100))

    sync_bn = False
    use_amp = True
    use_compile = False
    compile_params = {'mode': 'default'}

    use_ema_model = False
    ema_model_decay = 0.9999
    # MI355X engine: the whole iteration (forward .. zero_grad) as one replayed hipGraph (tools/scripts.py)
    use_step_graph = os.environ.get('SAICV_MAE_GRAPH', '1') == '1'
