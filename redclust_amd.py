"""Import shim: loads the package directory `redclust.jl_amd/` (not a valid identifier) as module
`redclust_amd`.  `import redclust_amd as rc` works wherever the repo root is on sys.path."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "redclust.jl_amd")
_spec = importlib.util.spec_from_file_location("redclust_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["redclust_amd"] = _mod
_spec.loader.exec_module(_mod)
