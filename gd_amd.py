"""Import shim: the package directory is named `3d-vlm-gd_amd/` (not a valid Python identifier),
so `import gd_amd` loads it under that module name.  Submodules: gd_amd._lib, gd_amd.ops, ..."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "3d-vlm-gd_amd")
_spec = importlib.util.spec_from_file_location(
    "gd_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gd_amd"] = _mod
_spec.loader.exec_module(_mod)
