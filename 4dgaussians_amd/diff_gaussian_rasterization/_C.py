"""Counterpart of the reference's pybind module `diff_gaussian_rasterization._C`
(RAST/ext.cpp:15-19, RAST/rasterize_points.cu:35-223): same three functions, same positional
arguments, same return tuples -- implemented over the C ABI of libgs4d.so (include/gs4d.h).
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _cabi

NUM_CHANNELS = 3


def _dev_f32(t: torch.Tensor, device) -> torch.Tensor:
    """contiguous float32 tensor on `device` (the reference calls .contiguous().data<float>())."""
    if t.numel() == 0:
        return t
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected a float32 tensor, got {t.dtype}")
    if t.device != device:
        raise RuntimeError(f"expected tensor on {device}, got {t.device}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, forward_only=False):
    """RasterizeGaussiansCUDA, RAST/rasterize_points.cu:35-117.

    forward_only (no reference counterpart): the returned state will not be handed to rasterize_gaussians_backward, so
    the blend leaves no segment checkpoints for it (GS4D_FORWARD_ONLY, include/gs4d.h)."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    L = _cabi.lib()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("rasterize_gaussians needs tensors on a HIP device; there is no CPU path")
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    # With P != 0 the forward writes every pixel (blend) and every radius (preprocess): no clearing fill.  With no
    # points the library returns at once (rasterize_points.cu:82) and the zero-initialised outputs are the result.
    # The three tensors escape to the caller, so each is an allocation of its own as in the reference
    # (rasterize_points.cu:68-70): views of one block would make autograd refuse in-place ops on them
    # (image.clamp_()) and would keep the whole block alive through any one of them.
    make = torch.zeros if P == 0 else torch.empty
    out_color = make((NUM_CHANNELS, H, W), dtype=torch.float32, device=device)
    out_depth = make((1, H, W), dtype=torch.float32, device=device)
    radii = make((P,), dtype=torch.int32, device=device)
    geom, binning, img = (_cabi.TensorAllocator(device) for _ in range(3))
    rendered = C.c_int(0)
    if P != 0:
        M = sh.size(1) if sh.numel() != 0 else 0
        keep = [_dev_f32(t, device) for t in (background, means3D, sh, colors, opacity, scales, rotations,
                                                cov3D_precomp, viewmatrix, projmatrix, campos)]
        bg_, means_, sh_, col_, op_, sc_, rot_, cov_, view_, proj_, cam_ = keep
        with torch.cuda.device(device):
            rc = L.gs4d_rasterize_forward(
                geom.callback, geom.ctx, binning.callback, binning.ctx, img.callback, img.ctx,
                C.c_int(P), C.c_int(int(degree)), C.c_int(M), _cabi.ptr(bg_), C.c_int(W), C.c_int(H),
                _cabi.ptr(means_), _cabi.ptr(sh_), _cabi.ptr(col_), _cabi.ptr(op_), _cabi.ptr(sc_),
                C.c_float(scale_modifier), _cabi.ptr(rot_), _cabi.ptr(cov_), _cabi.ptr(view_), _cabi.ptr(proj_),
                _cabi.ptr(cam_), C.c_float(tan_fovx), C.c_float(tan_fovy), C.c_int(int(bool(prefiltered))),
                _cabi.ptr(out_color), _cabi.ptr(out_depth), _cabi.ptr(radii),
                C.c_int(int(bool(debug)) | (2 if forward_only else 0)), _cabi.current_stream(device), C.byref(rendered))
        for a in (geom, binning, img):
            a.close()
        _cabi.check(rc, "rasterize_gaussians")
    else:
        for a in (geom, binning, img):
            a.close()
    return rendered.value, out_color, out_depth, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_gaussians_forward_only(*args):
    return rasterize_gaussians(*args, forward_only=True)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_depth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA, RAST/rasterize_points.cu:119-202."""
    L = _cabi.lib()
    device = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.numel() != 0 else 0
    (dL_dmeans3D, dL_dcolors, dL_ddepths, dL_dconic, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales,
     dL_drotations) = _cabi.zeros_group(device, (P, 3), (P, NUM_CHANNELS), (P, 1), (P, 2, 2), (P, 1), (P, 6),
                                        (P, M, 3), (P, 3), (P, 4))
    # dL_dmeans2D becomes viewspace_point_tensor.grad, which the densifier keeps per view (train.py:231-233): its own
    # allocation, so that it does not pin the ~76 P floats of the block above.  The per-Gaussian kernel writes all of
    # its 3 P elements whenever it runs (R > 0), so it only needs clearing when nothing was rendered.
    dL_dmeans2D = (torch.zeros if (P == 0 or int(R) == 0) else torch.empty)((P, 3), dtype=torch.float32, device=device)
    if P != 0:
        keep = [_dev_f32(t, device) for t in (background, means3D, sh, colors, scales, rotations, cov3D_precomp,
                                                viewmatrix, projmatrix, campos, dL_dout_color, dL_dout_depth)]
        bg_, means_, sh_, col_, sc_, rot_, cov_, view_, proj_, cam_, gcol_, gdep_ = keep
        radii_ = radii.contiguous()
        scratch = _cabi.TensorAllocator(device)
        with torch.cuda.device(device):
            rc = L.gs4d_rasterize_backward(
                scratch.callback, scratch.ctx, C.c_int(P), C.c_int(int(degree)), C.c_int(M), C.c_int(int(R)),
                _cabi.ptr(bg_), C.c_int(W), C.c_int(H), _cabi.ptr(means_), _cabi.ptr(sh_), _cabi.ptr(col_),
                _cabi.ptr(sc_), C.c_float(scale_modifier), _cabi.ptr(rot_), _cabi.ptr(cov_), _cabi.ptr(view_),
                _cabi.ptr(proj_), _cabi.ptr(cam_), C.c_float(tan_fovx), C.c_float(tan_fovy), _cabi.ptr(radii_),
                _cabi.ptr(geomBuffer), _cabi.ptr(binningBuffer), _cabi.ptr(imageBuffer), _cabi.ptr(gcol_),
                _cabi.ptr(gdep_), _cabi.ptr(dL_dmeans2D), _cabi.ptr(dL_dconic), _cabi.ptr(dL_dopacity),
                _cabi.ptr(dL_dcolors), _cabi.ptr(dL_ddepths), _cabi.ptr(dL_dmeans3D), _cabi.ptr(dL_dcov3D),
                _cabi.ptr(dL_dsh), _cabi.ptr(dL_dscales), _cabi.ptr(dL_drotations), C.c_int(int(bool(debug))),
                _cabi.current_stream(device))
        scratch.close()
        _cabi.check(rc, "rasterize_gaussians_backward")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, RAST/rasterize_points.cu:204-223."""
    L = _cabi.lib()
    device = means3D.device
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=device)
    if P != 0:
        means_, view_, proj_ = (_dev_f32(t, device) for t in (means3D, viewmatrix, projmatrix))
        with torch.cuda.device(device):
            rc = L.gs4d_mark_visible(C.c_int(P), _cabi.ptr(means_), _cabi.ptr(view_), _cabi.ptr(proj_),
                                     _cabi.ptr(present), _cabi.current_stream(device))
        _cabi.check(rc, "mark_visible")
    return present


def state_view(P, R, W, H, geomBuffer, binningBuffer, imageBuffer) -> dict:
    """Debug accessor (no reference counterpart): torch views of the arrays inside the opaque
    scratch chunks, for bit-exact comparison with the oracle."""
    L = _cabi.lib()
    v = _cabi.StateView()
    _cabi.check(L.gs4d_state_view_get(C.c_int(P), C.c_int(R), C.c_int(W), C.c_int(H), _cabi.ptr(geomBuffer),
                                      _cabi.ptr(binningBuffer), _cabi.ptr(imageBuffer), C.byref(v)), "state_view")
    gx, gy = (W + 15) // 16, (H + 15) // 16

    def view(buf, p, dtype, shape):
        n = 1
        for s in shape:
            n *= s
        if p is None or n == 0:
            return torch.zeros(shape, dtype=dtype, device=buf.device)
        off = p - buf.data_ptr()
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return buf[off:off + nbytes].view(dtype).view(shape)

    g, b, im = geomBuffer, binningBuffer, imageBuffer
    return dict(
        depths=view(g, v.depths, torch.float32, (P,)), clamped=view(g, v.clamped, torch.uint8, (P, 3)),
        means2D=view(g, v.means2D, torch.float32, (P, 2)), cov3D=view(g, v.cov3D, torch.float32, (P, 6)),
        conic_opacity=view(g, v.conic_opacity, torch.float32, (P, 4)), rgb=view(g, v.rgb, torch.float32, (P, 3)),
        tiles_touched=view(g, v.tiles_touched, torch.int32, (P,)), point_offsets=view(g, v.point_offsets, torch.int32, (P,)),
        keys_unsorted=view(b, v.keys_unsorted, torch.int64, (R,)), values_unsorted=view(b, v.values_unsorted, torch.int32, (R,)),
        keys_sorted=view(b, v.keys_sorted, torch.int64, (R,)), point_list=view(b, v.point_list, torch.int32, (R,)),
        ranges=view(im, v.ranges, torch.int32, (gx * gy, 2)), final_T=view(im, v.final_T, torch.float32, (H, W)),
        n_contrib=view(im, v.n_contrib, torch.int32, (H, W)))
