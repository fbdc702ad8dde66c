from .resnet import *
from .resnetforcifar import *
from .vit import *
