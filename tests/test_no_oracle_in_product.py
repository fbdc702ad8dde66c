"""The product package must never import, call or shell out to anything under oracle/, and must
not silently fall back to CPU."""
import os
import re

import pytest
import torch

from conftest import ROOT

PKG = os.path.join(ROOT, 'simpleaicv_pytorch_training_examples_amd')


def test_product_never_references_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f), errors='replace').read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M) or 'torch_oracle' in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_product_fails_loudly_on_cpu_tensors():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones
    model = backbones.resnet18cifar(num_classes=10)
    with pytest.raises(RuntimeError):
        model(torch.randn(2, 3, 32, 32))


def test_relative_imports_stay_inside_the_package():
    """Every `from .. import x` in the product package resolves inside it -- also the lazy ones inside functions that no
    CPU test executes (a `from .... import ops` one level too deep sat in load_state_dict unnoticed for a round)."""
    import ast
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if not f.endswith('.py'):
                continue
            path = os.path.join(dirpath, f)
            rel = os.path.relpath(path, os.path.dirname(PKG)).split(os.sep)
            depth = len(rel) - 1                       # packages above the module, counting the top-level package
            for node in ast.walk(ast.parse(open(path).read())):
                if isinstance(node, ast.ImportFrom) and node.level > depth:
                    bad.append(f'{os.path.relpath(path, PKG)}:{node.lineno} level {node.level} > depth {depth}')
    assert not bad, bad
