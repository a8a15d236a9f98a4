"""Import shim: the package directory is named ``toy-example-of-ilqr_amd`` (not a valid Python
identifier), so ``import cilqr_amd`` loads it through importlib and re-exports it."""
import importlib
import pathlib
import sys

_root = str(pathlib.Path(__file__).resolve().parent)
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("toy-example-of-ilqr_amd")
sys.modules[__name__] = _pkg
