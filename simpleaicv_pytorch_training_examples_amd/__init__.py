"""MI355X-native (gfx950) engine for the SimpleAICV DDP forward/backward hot path.

Sub-packages mirror the reference layout so that a reference `train_config.py` only changes
its import root:

    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones, losses
    from simpleaicv_pytorch_training_examples_amd.tools.utils import build_training_mode, ...

Kernels live in csrc/ (hand-written HIP for gfx950) behind the C-ABI in include/saicv_hip.h.
"""
__version__ = '0.1.0'

import os as _os

# r06 -- ROCm's "graph packet capture" (hipGraph replays as pre-recorded AQL packets, DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default in
# ROCm 7.2) breaks the SAM training step as a captured graph: work enqueued on the launch stream after a replay starts before the replay
# has finished, and replays of one graph between other work turn its outputs to garbage (with the switch at 0
# the captured loop equals the eager loop bit for bit and two prompt combinations alternate cleanly).  The HIP runtime reads the switch
# once, before the process's first HIP call -- so what matters is the environment at that moment.  GRAPH_PACKET_CAPTURE_OFF records
# what this package saw when it was imported; tools.interactive_segmentation_scripts allows the full captured SAM step only then.
# The other captured steps (ResNet / ViT / DETR / RetinaNet / MAE) are bit-exact against their eager loops either way and keep the
# runtime's default (packet capture saves 0.6 ms / 4.3 ms of host enqueue per ResNet-50 / DETR step, 0.04 / 0.19 ms of step time).
# A captured step of an N-rank job carries RCCL kernel nodes on a second stream and has never run on hardware here: it takes the
# conventional replay path too (one unknown fewer for 0.04 ms of a ResNet-50 step).  Only if HIP is not initialised yet -- later the
# runtime would not see the change.
import sys as _sys


def _hip_initialised():
    t = _sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:       # noqa: BLE001
        return False


if (int(_os.environ.get('WORLD_SIZE', '1') or 1) > 1 or _os.environ.get('SAICV_SAM_GRAPH') == '1') and not _hip_initialised():
    _os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
GRAPH_PACKET_CAPTURE_OFF = _os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0'
