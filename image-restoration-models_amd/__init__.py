"""MI355X-native hot path of leducthanhig/image-restoration-models.

The directory mirrors the reference's ``src/`` (configs, utils, restormer, dncnn,
rednet, ...) so the reference's scripts can be pointed at it (see
INTEGRATION.md); the compute runs in ``libirm_hip.so`` (csrc/, C ABI in
include/irm_hip.h).  Import as ``irm_amd`` through the shim at the repo root.
"""
import sys as _sys

__version__ = "0.1.0"

_SRC_MODULES = ("configs", "utils", "restormer", "dncnn", "rednet", "mair", "deblurganv2")


def install_as_src():
    """Bind this package's modules to the reference's top-level names
    (``import utils``, ``import restormer`` ...), i.e. what
    ``sys.path.append('src')`` gives the reference's scripts."""
    import importlib
    for name in _SRC_MODULES:
        _sys.modules[name] = importlib.import_module(f"{__name__}.{name}")
