from .detr import *
from .retinanet import *
from .fcos import *
from .dinov3_vit_retinanet import *
from .dinov3_vit_fcos import *
