from .vit_mae import *
