"""Import alias: the package directory is named `rl-gan_amd` (not a Python identifier), so
`import rl_gan_amd` (and `rl_gan_amd.<submodule>`) resolve to it through importlib.  Every
submodule is registered under both names so there is exactly one copy of each in the process."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("rl-gan_amd")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("rl-gan_amd."):
        sys.modules["rl_gan_amd." + _name[len("rl-gan_amd."):]] = _mod
sys.modules[__name__] = _pkg
