"""`import dfgpu` alias for the package directory `datafusion-upstream_amd/` (a hyphen cannot be written
in an import statement).  `dfgpu.x` and `datafusion-upstream_amd.x` are ONE module object: a finder maps every `dfgpu.*` import onto the
real package, so module-level state (e.g. device.py's registry of lent device memory) exists once whatever name a caller imports."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = "datafusion-upstream_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith("dfgpu."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len("dfgpu"):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules["dfgpu" + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
