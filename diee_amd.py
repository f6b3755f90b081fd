"""Import shim: `import diee_amd` == importlib.import_module("die-e_amd") (the package directory
name required by the project layout is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("die-e_amd")
sys.modules[__name__] = _pkg
