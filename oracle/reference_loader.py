"""
ORACLE (test infrastructure) -- loads the LIVE reference modules `diffusion.py` and `utils.py`
from /root/reference through a synthetic package object that skips the package `__init__`
(which imports a_unet / torchaudio, both absent here).

Only usable in the build container (the GPU box has no /root/reference): used to pin the
restated v-diffusion math in oracle/vdiffusion.py and to generate tests/golden fixtures.
"""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_PKG = "audio_diffusion_pytorch"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, _PKG, "diffusion.py"))


def load_reference():
    """Returns (diffusion_module, utils_module) of the live reference."""
    if not reference_available():
        raise RuntimeError("/root/reference is not present on this machine")
    name = "_ref_" + _PKG
    if name + ".diffusion" in sys.modules:
        return sys.modules[name + ".diffusion"], sys.modules[name + ".utils"]
    pkg = types.ModuleType(name)
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, _PKG)]
    sys.modules[name] = pkg
    mods = {}
    for sub in ("utils", "diffusion"):
        spec = importlib.util.spec_from_file_location(
            f"{name}.{sub}", os.path.join(REFERENCE_ROOT, _PKG, f"{sub}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{name}.{sub}"] = mod
        spec.loader.exec_module(mod)
        mods[sub] = mod
    return mods["diffusion"], mods["utils"]
