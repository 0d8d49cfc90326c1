"""Import alias: the package directory is `ipdm-pytorch_amd/` (not a valid Python identifier), so
`import ipdm_pytorch_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ipdm-pytorch_amd")
_spec = importlib.util.spec_from_file_location("ipdm_pytorch_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ipdm_pytorch_amd"] = _mod
_spec.loader.exec_module(_mod)
