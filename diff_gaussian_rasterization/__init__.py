"""Import-path shim: `import diff_gaussian_rasterization` (as gaussian_renderer/renderer.py:14 of
the reference does) resolves to the MI355X implementation in 4dgaussians_amd."""
import importlib as _il
import sys as _sys

_impl = _il.import_module("4dgaussians_amd.diff_gaussian_rasterization")
_C = _impl._C
_sys.modules[__name__ + "._C"] = _C
GaussianRasterizationSettings = _impl.GaussianRasterizationSettings
GaussianRasterizer = _impl.GaussianRasterizer
rasterize_gaussians = _impl.rasterize_gaussians
_RasterizeGaussians = _impl._RasterizeGaussians
cpu_deep_copy_tuple = _impl.cpu_deep_copy_tuple
