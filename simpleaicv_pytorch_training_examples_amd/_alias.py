"""Import aliasing for the drop-in surface: the reference's configs and entry scripts write
`from SimpleAICV.classification import backbones`, `from tools.utils import build_optimizer`, ....
The thin top-level packages `SimpleAICV/` and `tools/` of this repository call install() so that those names resolve
to the SAME module objects as `simpleaicv_pytorch_training_examples_amd.SimpleAICV...` / `.tools...` (one copy of
every class, whichever spelling imported it first)."""
import importlib
import importlib.abc
import importlib.util
import sys


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, alias, target):
        self.alias, self.target = alias, target

    def _real_name(self, name):
        if name == self.alias:
            return self.target
        if name.startswith(self.alias + '.'):
            return self.target + name[len(self.alias):]
        return None

    def find_spec(self, name, path=None, target=None):
        real = self._real_name(name)
        if real is None:
            return None
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(name, self)

    def create_module(self, spec):
        return importlib.import_module(self._real_name(spec.name))     # the one real module object

    def exec_module(self, module):
        pass


def install(alias, target):
    for f in sys.meta_path:
        if isinstance(f, _AliasFinder) and f.alias == alias:
            break
    else:
        sys.meta_path.insert(0, _AliasFinder(alias, target))
    sys.modules[alias] = importlib.import_module(target)
