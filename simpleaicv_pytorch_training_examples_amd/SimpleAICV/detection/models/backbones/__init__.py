from .detr_resnet import *
from .resnet import *
from .vit import *
from .dinov3vit import *
from .van import *
from .convformer import *
from .dinov3convnext import *
