from .detr_resnet import *
from .resnet import *
from .vit import *
