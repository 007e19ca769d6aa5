"""Import-path shim: `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:22 of the
reference) resolves to the MI355X implementation in 4dgaussians_amd."""
import importlib as _il
import sys as _sys

_C = _il.import_module("4dgaussians_amd.simple_knn._C")
_sys.modules[__name__ + "._C"] = _C
