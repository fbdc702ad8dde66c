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
# ROCm 7.2) gave wrong results in two independent captured steps here:
#   * the SAM training step: work enqueued on the launch stream after a replay starts before the replay has finished, and replays of
#     one graph between other work turn its outputs to garbage;
#   * a deterministic-mode ResNet-50 b256 step once the weight-gradient partials workspace (reused by all 54 layers of a step) is no
#     longer cleared in front of every use (csrc/det.hip): from the SECOND replay on the weights differ from the eager loop's, in 5 of 5
#     processes (scripts/probes/bench_repro_probe.py) -- with the switch at 0 the same graph equals the eager loop bit for bit, 5 of 5.
# Same-box A/B (profiles/r06_nt_experiments.md section 6): ResNet-50 20.42 / 20.41 ms, ViT-B 39.02 / 39.02 ms with it on / off, DETR
# +0.19 ms -- nothing to keep it for.  So EVERY captured step takes the conventional replay path: the switch is set to 0 here unless
# the environment already sets it.  The HIP runtime reads it once, before the process's first HIP call, so this only works if HIP is not
# initialised yet; GRAPH_PACKET_CAPTURE_OFF records what the runtime will have seen.  With it on (a process that initialised HIP before
# importing this package) engine.StepGraph warns once and runs every step eagerly.
import sys as _sys


def _hip_initialised():
    t = _sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:       # noqa: BLE001
        return False


_HIP_WAS_UP = _hip_initialised()
if not _HIP_WAS_UP:
    _os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
GRAPH_PACKET_CAPTURE_OFF = _os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0'
