from .detr import *
