from .detr import *
from .retinanet import *
from .fcos import *
