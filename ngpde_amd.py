"""Import shim: `import ngpde_amd` loads the package in ./neuralgraphpde.jl_amd/ (a directory whose
name is not a valid Python identifier) under the module name `ngpde_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neuralgraphpde.jl_amd")
_spec = importlib.util.spec_from_file_location("ngpde_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ngpde_amd"] = _mod
_spec.loader.exec_module(_mod)
