"""Counterpart of `simple_knn._C` (submodules/simple-knn/ext.cpp, spatial.cu:15-26)."""
import ctypes as C

import torch

from .. import _cabi


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """(P,3) float32 device tensor -> (P,) mean squared distance to the 3 nearest neighbours."""
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 needs a tensor on a HIP device; there is no CPU path")
    if points.dtype != torch.float32 or points.ndimension() != 2 or points.size(1) != 3:
        raise RuntimeError("points must be a float32 tensor of shape (P, 3)")
    L = _cabi.lib()
    pts = points.contiguous()
    P = pts.size(0)
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    if P:
        scratch = _cabi.TensorAllocator(pts.device)
        with torch.cuda.device(pts.device):
            rc = L.gs4d_knn_mean_dist2(scratch.callback, scratch.ctx, C.c_int(P), _cabi.ptr(pts), _cabi.ptr(out),
                                       _cabi.current_stream(pts.device))
        scratch.close()
        _cabi.check(rc, "distCUDA2")
    return out
