"""Import shim: ``import irm_amd`` loads the package that lives in
``image-restoration-models_amd/`` (the directory name the build contract asks
for is not a valid Python identifier, so it is bound to the name ``irm_amd``).

After ``import irm_amd`` the usual submodule imports work:
``from irm_amd import restormer, utils, configs``.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image-restoration-models_amd")


def _load():
    spec = importlib.util.spec_from_file_location(
        "irm_amd", os.path.join(_PKG_DIR, "__init__.py"),
        submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["irm_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


_load()
