"""`import tools.utils / tools.scripts / tools.path ...` as the reference's entry scripts spell it:
resolves to simpleaicv_pytorch_training_examples_amd.tools (same module objects)."""
from simpleaicv_pytorch_training_examples_amd._alias import install as _install

_install(__name__, 'simpleaicv_pytorch_training_examples_amd.tools')
