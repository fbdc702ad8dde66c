from .resnet import *
from .resnetforcifar import *
from .vit import *
from .darknet import *
from .van import *
from .convformer import *
