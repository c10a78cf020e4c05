"""Import alias: the package directory is `radix-sort_amd/` (not a Python identifier); importing this
module registers that directory as the package `radix_sort_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "radix-sort_amd")
_spec = importlib.util.spec_from_file_location(
    "radix_sort_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["radix_sort_amd"] = _mod
_spec.loader.exec_module(_mod)
