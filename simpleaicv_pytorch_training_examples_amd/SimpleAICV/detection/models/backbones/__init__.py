from .detr_resnet import *
