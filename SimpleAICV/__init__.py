"""`import SimpleAICV...` as the reference's train_config.py files spell it: every submodule resolves to
the MI355X implementation in simpleaicv_pytorch_training_examples_amd.SimpleAICV (same module objects)."""
from simpleaicv_pytorch_training_examples_amd._alias import install as _install

_install(__name__, 'simpleaicv_pytorch_training_examples_amd.SimpleAICV')
