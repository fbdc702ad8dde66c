from .sam import *
