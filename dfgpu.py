"""`import dfgpu` alias for the package directory `datafusion-upstream_amd/` (a hyphen cannot be written
in an import statement)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("datafusion-upstream_amd")
sys.modules[__name__] = _pkg
