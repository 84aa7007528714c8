"""Import alias: the package directory is named `rl-gan_amd` (not a Python identifier), so
`import rl_gan_amd` resolves to it through importlib."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("rl-gan_amd")
sys.modules[__name__] = _pkg
