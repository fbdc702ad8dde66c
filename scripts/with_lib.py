"""Run a script against a VARIANT library (scripts/build_variant_lib.py):   python scripts/with_lib.py <tag> <script.py> [args ...]
The package's loader is pointed at libsaicv_hip_<tag>.so before anything calls into it; tag "-" = the product library."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, script = sys.argv[1], sys.argv[2]
if tag != '-':
    from simpleaicv_pytorch_training_examples_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f'libsaicv_hip_{tag}.so')
    assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH
sys.argv = [script] + sys.argv[3:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name='__main__')
