"""ctypes/numpy front-end of oracle/rasterizer_ref.c -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It mirrors the call sequence of the reference's Rasterizer::forward /
::backward (RAST/cuda_rasterizer/rasterizer_impl.cu:198-444) and returns every
intermediate as a numpy array so the HIP path can be compared stage by stage.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_rasterizer.so")
_lib = None


def build(force: bool = False) -> None:
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rasterizer_ref.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_rasterizer.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gs4d_ref_preprocess.restype = C.c_int64
        _lib.gs4d_ref_higher_msb.restype = C.c_uint32
    return _lib


def _f(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _c32(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a if a.size else None


def higher_msb(n: int) -> int:
    return int(lib().gs4d_ref_higher_msb(C.c_uint32(n)))


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    means3D = _c32(means3D)
    P = 0 if means3D is None else means3D.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    if P:
        lib().gs4d_ref_mark_visible(P, _f(means3D), _f(_c32(viewmatrix)), _f(_c32(projmatrix)), _p(out))
    return out.astype(bool)


def forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
            viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos,
            prefiltered=False) -> dict:
    """Same argument order as _C.rasterize_gaussians (RAST/diff_gaussian_rasterization/__init__.py:60-80)."""
    L = lib()
    bg = _c32(bg); means3D = _c32(means3D); colors_precomp = _c32(colors_precomp); opacities = _c32(opacities)
    scales = _c32(scales); rotations = _c32(rotations); cov3D_precomp = _c32(cov3D_precomp)
    viewmatrix = _c32(viewmatrix); projmatrix = _c32(projmatrix); sh = _c32(sh); campos = _c32(campos)
    H, W = int(image_height), int(image_width)
    P = 0 if means3D is None else int(means3D.shape[0])
    M = 0 if sh is None else int(sh.shape[1])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r = dict(P=P, M=M, W=W, H=H, D=int(degree))
    r["radii"] = np.zeros(P, np.int32)
    r["means2D"] = np.zeros((P, 2), np.float32)
    r["depths"] = np.zeros(P, np.float32)
    r["cov3D"] = np.zeros((P, 6), np.float32)
    r["rgb"] = np.zeros((P, 3), np.float32)
    r["conic_opacity"] = np.zeros((P, 4), np.float32)
    r["clamped"] = np.zeros((P, 3), np.uint8)
    r["tiles_touched"] = np.zeros(P, np.uint32)
    r["point_offsets"] = np.zeros(P, np.uint32)
    r["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    r["color"] = np.zeros((3, H, W), np.float32)
    r["depth"] = np.zeros((1, H, W), np.float32)
    r["final_T"] = np.zeros((H, W), np.float32)
    r["n_contrib"] = np.zeros((H, W), np.uint32)
    r["margin"] = np.full((H, W), np.float32(3.4e38), np.float32)
    R = 0
    if P:
        R = int(L.gs4d_ref_preprocess(
            P, int(degree), M, _f(means3D), _f(scales), C.c_float(scale_modifier), _f(rotations), _f(opacities),
            _f(sh), _f(cov3D_precomp), _f(colors_precomp), _f(viewmatrix), _f(projmatrix), _f(campos),
            W, H, C.c_float(tanfovx), C.c_float(tanfovy), int(bool(prefiltered)),
            _p(r["radii"]), _p(r["means2D"]), _p(r["depths"]), _p(r["cov3D"]), _p(r["rgb"]),
            _p(r["conic_opacity"]), _p(r["clamped"]), _p(r["tiles_touched"]), _p(r["point_offsets"])))
        if R < 0:
            raise RuntimeError("Point is filtered although prefiltered is set")
    r["num_rendered"] = R
    r["keys_unsorted"] = np.zeros(R, np.uint64)
    r["values_unsorted"] = np.zeros(R, np.uint32)
    r["keys_sorted"] = np.zeros(R, np.uint64)
    r["point_list"] = np.zeros(R, np.uint32)
    if P:
        L.gs4d_ref_binning(P, C.c_int64(R), W, H, _p(r["means2D"]), _p(r["depths"]), _p(r["point_offsets"]),
                           _p(r["radii"]), _p(r["keys_unsorted"]), _p(r["values_unsorted"]),
                           _p(r["keys_sorted"]), _p(r["point_list"]), _p(r["ranges"]))
        feat = colors_precomp if colors_precomp is not None else r["rgb"]
        L.gs4d_ref_render_forward(W, H, _p(r["ranges"]), _p(r["point_list"]), _p(r["means2D"]), _f(feat),
                                  _p(r["depths"]), _p(r["conic_opacity"]), _f(bg), _p(r["color"]), _p(r["depth"]),
                                  _p(r["final_T"]), _p(r["n_contrib"]), _p(r["margin"]))
    r["_inputs"] = dict(bg=bg, means3D=means3D, colors_precomp=colors_precomp, opacities=opacities, scales=scales,
                        rotations=rotations, scale_modifier=float(scale_modifier), cov3D_precomp=cov3D_precomp,
                        viewmatrix=viewmatrix, projmatrix=projmatrix, tanfovx=float(tanfovx), tanfovy=float(tanfovy),
                        sh=sh, campos=campos)
    return r


def backward(fwd: dict, dL_dout_color, dL_dout_depth, float_accum: bool = False) -> dict:
    """Mirrors Rasterizer::backward (rasterizer_impl.cu:343-444) on a forward() result."""
    L = lib()
    i = fwd["_inputs"]
    P, W, H, M, D = fwd["P"], fwd["W"], fwd["H"], fwd["M"], fwd["D"]
    dL_dout_color = _c32(dL_dout_color).reshape(3, H, W)
    dL_dout_depth = _c32(dL_dout_depth).reshape(H, W)
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
        dL_ddepths=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
        dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
        dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32))
    if not P:
        return g
    feat = i["colors_precomp"] if i["colors_precomp"] is not None else fwd["rgb"]
    L.gs4d_ref_render_backward(
        P, W, H, _p(fwd["ranges"]), _p(fwd["point_list"]), _f(i["bg"]), _p(fwd["means2D"]),
        _p(fwd["conic_opacity"]), _f(feat), _p(fwd["depths"]), _p(fwd["final_T"]), _p(fwd["n_contrib"]),
        _f(dL_dout_color), _f(dL_dout_depth), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]),
        _p(g["dL_dcolors"]), _p(g["dL_ddepths"]), int(bool(float_accum)))
    cov3D = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else fwd["cov3D"]
    L.gs4d_ref_preprocess_backward(
        P, D, M, _f(i["means3D"]), _p(fwd["radii"]), _f(i["sh"]), _p(fwd["clamped"]), _f(i["scales"]),
        _f(i["rotations"]), C.c_float(i["scale_modifier"]), _f(cov3D), _f(i["viewmatrix"]), _f(i["projmatrix"]),
        W, H, C.c_float(i["tanfovx"]), C.c_float(i["tanfovy"]), _f(i["campos"]),
        _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcolors"]), _p(g["dL_ddepths"]),
        _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g
