"""Child process of tests/test_zz_gpu_trajectories.py::test_sam_step_graphs_of_two_prompt_combinations_alternate_with_graph_packet_capture_off
(the parent puts DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 into the environment, which the HIP runtime reads before its first call): the tiny SAM
loop with the reference config's point-only / box-only draw, 18 iterations, eagerly and with the captured step; prints one JSON line."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402

import simpleaicv_pytorch_training_examples_amd as pkg  # noqa: E402
import test_zz_gpu_trajectories as T  # noqa: E402
from simpleaicv_pytorch_training_examples_amd import engine, ops  # noqa: E402


class MP:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


ops.set_deterministic(True)
made = []
orig_init = engine.StepGraph.__init__


def spy_init(self, *a, **k):
    orig_init(self, *a, **k)
    made.append(self)


engine.StepGraph.__init__ = spy_init
_, eager, _, p_eager = T._run_sam_tiny('iters', MP(), False, True, steps_override=18, mixed=True)
_, graph, _, p_graph = T._run_sam_tiny('iters', MP(), True, True, steps_override=18, mixed=True)
print(json.dumps({'packet_capture_off': pkg.GRAPH_PACKET_CAPTURE_OFF, 'graphs': len(made), 'captured': sum(g.graph is not None for g in made),
                  'eager': eager, 'graph': graph,
                  'params_equal': bool(torch.equal(p_eager, p_graph))}))
