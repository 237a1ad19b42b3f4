"""Import shim: loads the package that lives in ``discretepomp.jl_amd/`` (a directory name
Python cannot import directly because of the dot) under the module name ``dpomp_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "discretepomp.jl_amd")
_spec = importlib.util.spec_from_file_location("dpomp_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dpomp_amd"] = _mod
_spec.loader.exec_module(_mod)
