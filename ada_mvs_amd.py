"""Import shim: `import ada_mvs_amd` -> the package in ./ada-mvs_amd/ (whose
directory name is not a legal Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ada-mvs_amd")
_spec = importlib.util.spec_from_file_location(
    "ada_mvs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ada_mvs_amd"] = _mod
_spec.loader.exec_module(_mod)
