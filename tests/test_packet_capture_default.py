"""Host-side: importing the package before the process's first HIP call switches ROCm's graph packet capture off for every captured
step (simpleaicv_pytorch_training_examples_amd/__init__.py), unless the environment already decides."""
import json
import os
import subprocess
import sys

from conftest import ROOT

CODE = ("import os, json, simpleaicv_pytorch_training_examples_amd as p; "
        "print(json.dumps([os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), p.GRAPH_PACKET_CAPTURE_OFF]))")


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k != 'DEBUG_CLR_GRAPH_PACKET_CAPTURE'}
    env.update(extra, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, '-c', CODE], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_import_switches_graph_packet_capture_off():
    assert _run({}) == ['0', True]


def test_an_explicit_setting_is_kept_and_reported():
    assert _run({'DEBUG_CLR_GRAPH_PACKET_CAPTURE': '1'}) == ['1', False]
