"""MI355X-native (gfx950) engine for the SimpleAICV DDP forward/backward hot path.

Sub-packages mirror the reference layout so that a reference `train_config.py` only changes
its import root:

    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones, losses
    from simpleaicv_pytorch_training_examples_amd.tools.utils import build_training_mode, ...

Kernels live in csrc/ (hand-written HIP for gfx950) behind the C-ABI in include/saicv_hip.h.
"""
__version__ = '0.1.0'
