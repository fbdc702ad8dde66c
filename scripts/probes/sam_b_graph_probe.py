"""Debug aid (r06; the SAICV_SAM_GRAPH_* / SAICV_GRAPH_COPY experiment switches it was run with are gone from the product again -- what remains
useful is DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 / 1 in the environment): the full SAM-B step of the reference config through bench.loop_workload, per-iteration log lines printed.
    python scripts/probes/sam_b_graph_probe.py <graph 0|1> <p_point> <steps> [batch]"""
import argparse
import logging
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

graph, p_point, steps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 8
args = argparse.Namespace(batch=batch, steps=steps, warmup=0)
device = torch.device('cuda:0')
torch.cuda.set_device(device)
run, model, scaler, state, info, b, size = bench.loop_workload('sam_b', args, 1, 0, device, bool(graph))
config = bench._CONFIGS['sam_b']
config.prompt_probs = {'prompt_point': p_point, 'prompt_box': 1.0 - p_point, 'prompt_mask': 0.}
if os.environ.get('PROBE_DECODER_ITERS'):
    config.decoder_iters = int(os.environ['PROBE_DECODER_ITERS'])
config.print_interval = 1
lg = logging.getLogger('saicv_bench')
lg.propagate = False
lg.setLevel(logging.INFO)
h = logging.StreamHandler(sys.stdout)
lg.addHandler(h)
config.local_rank, config.total_rank = 0, 0
run(steps)
torch.cuda.synchronize()
graphs = getattr(config, '_saicv_step_graphs', None) or {}
print('graphs', [(k[2], k[3], g.graph is not None, g.replays) for k, g in graphs.items()], 'scale', scaler.get_scale() if scaler is not None else None)
