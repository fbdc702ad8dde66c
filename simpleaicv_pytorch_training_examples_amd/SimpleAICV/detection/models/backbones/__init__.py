from .detr_resnet import *
from .resnet import *
from .vit import *
from .dinov3vit import *
