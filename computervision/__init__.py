"""Namespace shim: makes the directory ``computervision.pytorch_amd/`` importable as the module
``computervision.pytorch_amd`` (a dot cannot appear in a plain package directory name)."""
import importlib.util
import os
import sys

_real = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "computervision.pytorch_amd")
_name = __name__ + ".pytorch_amd"
if _name not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_name, os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_name] = _mod
    _spec.loader.exec_module(_mod)
pytorch_amd = sys.modules[_name]
