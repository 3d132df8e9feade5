"""Importable alias for the package directory ``se-unet-airseg_amd`` (not a valid identifier).

``import seunet_amd`` and ``from seunet_amd.<submodule> import ...`` resolve to the one set of module
objects of the real package (no second copy is ever loaded)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = "se-unet-airseg_amd"
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
