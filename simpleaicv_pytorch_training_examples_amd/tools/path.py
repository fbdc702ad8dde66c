"""Dataset roots the reference's configs import (reference tools/path.py).  The benchmark configs in this repository
use synthetic datasets; these names exist so an unmodified reference train_config.py imports cleanly.  Override with the
environment variable SAICV_DATA_ROOT."""
import os

_ROOT = os.environ.get('SAICV_DATA_ROOT', '/root/autodl-tmp')

# classification
CIFAR10_path = os.path.join(_ROOT, 'CIFAR10')
CIFAR100_path = os.path.join(_ROOT, 'CIFAR100')
ILSVRC2012_path = os.path.join(_ROOT, 'ILSVRC2012')
ImageNet21K_path = os.path.join(_ROOT, 'ImageNet21K')
# detection
COCO2017_path = os.path.join(_ROOT, 'COCO2017')
Objects365_path = os.path.join(_ROOT, 'objects365_2020')
VOCdataset_path = os.path.join(_ROOT, 'VOCdataset')
# interactive segmentation
interactive_segmentation_dataset_path = os.path.join(_ROOT, 'interactive_segmentation_dataset')
