"""Importable alias of the package directory `prob-unet-climate-downscaling_amd/` (its name is not a Python
identifier).  `import probunet_amd as pa; pa.ProbabilisticUNet(...)`."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "prob-unet-climate-downscaling_amd")
_spec = _u.spec_from_file_location("probunet_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["probunet_amd"] = _mod
_spec.loader.exec_module(_mod)
