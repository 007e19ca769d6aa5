"""ctypes/numpy front-end of oracle/knn_ref.c -- TEST INFRASTRUCTURE (see that file's header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_knn.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "knn_ref.c")):
            subprocess.check_call(["make", "-C", _HERE, "liboracle_knn.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def brute(points: np.ndarray) -> np.ndarray:
    pts = np.ascontiguousarray(points, np.float32)
    out = np.zeros(pts.shape[0], np.float32)
    lib().gs4d_ref_knn_brute(pts.shape[0], pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def morton(points: np.ndarray):
    pts = np.ascontiguousarray(points, np.float32)
    P = pts.shape[0]
    out = np.zeros(P, np.float32)
    codes = np.zeros(P, np.uint32)
    order = np.zeros(P, np.uint32)
    lib().gs4d_ref_knn_morton(P, pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                              codes.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p))
    return out, codes, order


def knn_query(xyz: np.ndarray, new_xyz: np.ndarray, nsample: int):
    """knnquery_cuda_kernel (one batch segment): -> (idx int32 [m, nsample], dist2 float32 [m, nsample])."""
    assert 1 <= nsample <= 100
    a = np.ascontiguousarray(xyz, np.float32)
    b = np.ascontiguousarray(new_xyz, np.float32)
    idx = np.zeros((b.shape[0], nsample), np.int32)
    d2 = np.zeros((b.shape[0], nsample), np.float32)
    lib().gs4d_ref_knn_query(a.shape[0], a.ctypes.data_as(C.c_void_p), b.shape[0], b.ctypes.data_as(C.c_void_p),
                             nsample, idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p))
    return idx, d2
