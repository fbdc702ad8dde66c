from .detr import *
from .retinanet import *
