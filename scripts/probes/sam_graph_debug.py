"""Debug aid (r06; the SAICV_SAM_GRAPH_* / SAICV_GRAPH_COPY experiment switches it was run with are gone from the product again -- what remains
useful is DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 / 1 in the environment): the tiny SAM loop of tests/test_zz_gpu_trajectories.py with the step graph on, parameter / gradient norms printed
around every StepGraph call."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_zz_gpu_trajectories as T  # noqa: E402
from simpleaicv_pytorch_training_examples_amd import engine, ops  # noqa: E402


class MP:
    def __init__(self):
        self.undo = []

    def setattr(self, obj, name, val):
        self.undo.append((obj, name, getattr(obj, name)))
        setattr(obj, name, val)


regime = sys.argv[1] if len(sys.argv) > 1 else 'all'
ops.set_deterministic(True)
orig = engine.StepGraph.__call__
arena = {}


def spy(self, *inputs):
    out = orig(self, *inputs)
    if os.environ.get('NOSYNC'):
        return out
    torch.cuda.synchronize()
    a = arena.get('a')
    msg = f'call {self.calls} replays {self.replays} out {[round(float(v), 5) for v in out.tolist()]}'
    if a is not None:
        msg += f' |param| {float(a.flat_param.norm()):.6f} |grad| {float(a.flat_grad.norm()):.6f} finite {bool(torch.isfinite(a.flat_param).all())}'
    print(msg, flush=True)
    return out


engine.StepGraph.__call__ = spy
orig_arena = engine._arena_of


def arena_spy(m):
    a = orig_arena(m)
    arena['a'] = a
    return a


engine._arena_of = arena_spy
for graph in (False, True):
    print('==== step graph', graph, flush=True)
    fx, got, avg, p = T._run_sam_tiny(regime, MP(), graph, True)
    print('losses', [round(g, 5) for g in got], 'param norm', float(p.norm()), flush=True)
