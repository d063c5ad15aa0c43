"""Importable alias for the package directory `crossmodal-contrastive-learning_amd/` (a hyphen is
not a legal character in an `import` statement).  `import crossclr_amd` yields that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("crossmodal-contrastive-learning_amd")
sys.modules[__name__] = _pkg
