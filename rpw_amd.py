"""Importable alias of the package directory `ros2-recursive-patchwork-implementation_amd`
(its name, fixed by the build contract, is not a valid Python identifier)."""
import importlib as _il
import sys as _sys

_pkg = _il.import_module("ros2-recursive-patchwork-implementation_amd")
workloads = _il.import_module("ros2-recursive-patchwork-implementation_amd.workloads")
_sys.modules[__name__ + ".workloads"] = workloads
globals().update({k: v for k, v in vars(_pkg).items() if not k.startswith("__")})
