import os
import sys

# ROCm's graph packet capture off for every captured step (package __init__.py): the HIP runtime reads the switch at
# its first call, so it is set before torch is imported (the package sets the same default when a test module imports it)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import pytest  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)


def rel_err(a, b):
    """max |a-b| / max|b|  (scale-relative error used for every parity check)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture
def deterministic():
    """The engine's deterministic mode (ops.set_deterministic, what tools.utils.set_seed() turns on -- reference tools/utils.py:95-107)
    for the duration of one test: ordered reductions instead of fp32 atomics, bit-reproducible run to run."""
    from simpleaicv_pytorch_training_examples_amd import ops
    prev = ops.set_deterministic(True)
    yield
    ops.set_deterministic(prev)
